"""Rotated-box overlap / IoU / NMS of the 3-D detection branch.

CPU: the restatement oracle/iou3d_oracle.py against tests/golden/iou3d.npz — outputs of the REFERENCE's own device functions compiled
for the host (tests/golden/make_iou3d_golden.py) — and, when /root/reference is present, against a fresh build of them; the host-side
mirror of iou3d_utils.py on the ABI emulator.  GPU (-m gpu): the HIP kernels against the same goldens, against properties that hold at
any size (IoU(a, a) = 1, symmetry, overlap <= min area, NMS idempotence, kept boxes mutually below the threshold) and at 4 096 boxes."""
import os

import numpy as np
import pytest
import torch

import conftest
from oracle import iou3d_oracle as io

GOLD = np.load(os.path.join(conftest.GOLDEN, "iou3d.npz"))


def test_oracle_matches_reference_device_functions():
    ov = io.pairwise(GOLD["pair_a"], GOLD["pair_b"], False)
    iou = io.pairwise(GOLD["pair_a"], GOLD["pair_b"], True)
    assert np.array_equal(ov, GOLD["overlap"]) and np.array_equal(iou, GOLD["iou"])           # bit-exact (same fp32 operation order)
    for thr in (0.1, 0.5):
        for rot in (1, 0):
            assert np.array_equal(io.nms(GOLD["nms_small_boxes"], thr, bool(rot)), GOLD[f"nms_small_t{thr}_r{rot}"])


def test_golden_regenerates_from_the_reference_sources():
    """Only where /root/reference exists (the build container): rebuild the reference's device functions and compare."""
    from oracle import ref_import
    if not ref_import.reference_available():
        pytest.skip("reference sources not present on this machine")
    import ctypes
    import subprocess
    so = subprocess.check_output(["bash", os.path.join(conftest.ROOT, "oracle", "build_ref_iou3d.sh")]).decode().strip().splitlines()[-1]
    lib = ctypes.CDLL(so)
    a, b = np.ascontiguousarray(GOLD["pair_a"]), np.ascontiguousarray(GOLD["pair_b"])
    out = np.zeros((len(a), len(b)), np.float32)
    fp = lambda x: x.ctypes.data_as(ctypes.c_void_p)
    lib.ref_boxes_iou_bev(len(a), fp(a), len(b), fp(b), fp(out))
    assert np.array_equal(out, GOLD["iou"])


def test_host_mirror_on_emulator(emulated):
    """iou3d.py (the reference's iou3d_utils API) through the ABI emulator: score sorting, pre / post limits, index mapping."""
    import mtt_amd
    bx = torch.from_numpy(GOLD["nms_small_boxes"])
    g = torch.Generator().manual_seed(3)
    perm = torch.randperm(len(bx), generator=g)
    scores = torch.empty(len(bx))
    scores[perm] = torch.linspace(1.0, 0.0, len(bx))                # box perm[i] has the i-th highest score
    for rot, fn in ((1, mtt_amd.iou3d.nms_gpu), (0, mtt_amd.iou3d.nms_normal_gpu)):
        want = io.nms(bx[perm].numpy(), 0.1, bool(rot))
        got = fn(bx, scores, 0.1)
        assert torch.equal(got, perm[torch.from_numpy(want)])
    got = mtt_amd.iou3d.nms_gpu(bx, scores, 0.5, pre_maxsize=30, post_max_size=7)
    want = io.nms(bx[perm[:30]].numpy(), 0.5, True)[:7]
    assert torch.equal(got, perm[:30][torch.from_numpy(want)])
    iou = mtt_amd.iou3d.boxes_iou_bev(torch.from_numpy(GOLD["pair_a"][:6]), torch.from_numpy(GOLD["pair_b"][:5]))
    assert np.array_equal(iou.numpy(), GOLD["iou"][:6, :5])


def _rand_boxes(n, spread, seed):
    rng = np.random.default_rng(seed)
    c = rng.uniform(-spread, spread, (n, 2))
    wh = rng.uniform(0.5, 6.0, (n, 2))
    return torch.from_numpy(np.concatenate([c - wh / 2, c + wh / 2, rng.uniform(-np.pi, np.pi, (n, 1))], 1).astype(np.float32))


@pytest.mark.gpu
def test_hip_pairwise_matches_reference_golden():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import mtt_amd
    a, b = torch.from_numpy(GOLD["pair_a"]).cuda(), torch.from_numpy(GOLD["pair_b"]).cuda()
    ov = mtt_amd.iou3d.boxes_overlap_bev(a, b).cpu().numpy()
    iou = mtt_amd.iou3d.boxes_iou_bev(a, b).cpu().numpy()
    # device sinf / cosf / atan2f differ from glibc's by ulps: areas agree to ~1e-6 of the box scale, not bit for bit
    assert np.abs(ov - GOLD["overlap"]).max() < 2e-5 * max(1.0, float(GOLD["overlap"].max())), np.abs(ov - GOLD["overlap"]).max()
    assert np.abs(iou - GOLD["iou"]).max() < 2e-5, np.abs(iou - GOLD["iou"]).max()


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["small", "large"])
def test_hip_nms_matches_reference_golden(tag):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import mtt_amd
    bx = torch.from_numpy(GOLD[f"nms_{tag}_boxes"]).cuda()
    scores = torch.linspace(1.0, 0.0, len(bx)).cuda()               # already in score order, like the fixtures
    for thr in (0.1, 0.5):
        for rot, fn in ((1, mtt_amd.iou3d.nms_gpu), (0, mtt_amd.iou3d.nms_normal_gpu)):
            got = fn(bx, scores, thr).cpu().numpy()
            want = GOLD[f"nms_{tag}_t{thr}_r{rot}"]
            assert np.array_equal(got, want), (tag, thr, rot, len(got), len(want))


@pytest.mark.gpu
def test_hip_nms_4096_boxes_matches_reference_golden():
    """4 096 boxes (64 x 64 tiles of the suppression matrix, the device-side greedy pass over 64 mask words per row): kept sets equal to
    the reference's device functions compiled for the host (tests/golden/make_iou3d_golden.py; boxes regenerate from the stored seed)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import mtt_amd
    from tests.golden.make_iou3d_golden import boxes
    seed, n = (int(v) for v in GOLD["nms_4096_seed"])
    bx = torch.from_numpy(boxes(np.random.default_rng(seed), n, 0.9 * n ** 0.5)).cuda()
    scores = torch.linspace(1.0, 0.0, n).cuda()
    for rot, fn in ((1, mtt_amd.iou3d.nms_gpu), (0, mtt_amd.iou3d.nms_normal_gpu)):
        got = fn(bx, scores, 0.3).cpu().numpy()
        want = GOLD[f"nms_4096_t0.3_r{rot}"]
        assert np.array_equal(got, want.astype(got.dtype)), (rot, len(got), len(want))


@pytest.mark.gpu
def test_hip_properties_at_scale():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import mtt_amd
    bx = _rand_boxes(4096, 60.0, 11).cuda()
    iou = mtt_amd.iou3d.boxes_iou_bev(bx[:512], bx[:512])
    assert float((iou.diagonal() - 1).abs().max()) < 1e-4                                   # IoU(a, a) = 1
    assert float((iou - iou.t()).abs().max()) < 1e-4                                        # symmetry
    ov = mtt_amd.iou3d.boxes_overlap_bev(bx[:512], bx[512:1024])
    area = (bx[:, 2] - bx[:, 0]) * (bx[:, 3] - bx[:, 1])
    assert bool((ov <= torch.minimum(area[:512, None], area[None, 512:1024]) * (1 + 1e-4) + 1e-4).all()) and bool((ov >= 0).all())
    scores = torch.rand(4096, generator=torch.Generator().manual_seed(1)).cuda()
    for fn in (mtt_amd.iou3d.nms_gpu, mtt_amd.iou3d.nms_normal_gpu):
        keep = fn(bx, scores, 0.3)
        assert 0 < len(keep) <= 4096 and bool((scores[keep][:-1] >= scores[keep][1:]).all())   # score order
        again = fn(bx[keep], scores[keep], 0.3)
        assert torch.equal(again, torch.arange(len(keep), device=keep.device))               # idempotent: nothing left to suppress
    keep = mtt_amd.iou3d.nms_gpu(bx, scores, 0.3)
    kk = mtt_amd.iou3d.boxes_iou_bev(bx[keep], bx[keep])
    kk.fill_diagonal_(0)
    assert float(kk.max()) <= 0.3 + 1e-4                                                     # kept boxes are mutually below the threshold
