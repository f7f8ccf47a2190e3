"""Rotated-box NMS / pairwise IoU on the MI355X next to the reference's own device functions compiled for the host (oracle/_ref, built
by oracle/build_ref_iou3d.sh in the build container; one host core).  Prints microseconds per call and pairs / s."""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import mtt_amd  # noqa: E402


def boxes(n, spread, seed):
    rng = np.random.default_rng(seed)
    c = rng.uniform(-spread, spread, (n, 2))
    wh = rng.uniform(0.5, 6.0, (n, 2))
    return np.concatenate([c - wh / 2, c + wh / 2, rng.uniform(-np.pi, np.pi, (n, 1))], 1).astype(np.float32)


ref = None
so = os.path.join(ROOT, "oracle", "_ref", "libiou3d_ref.so")
if os.path.exists(so):
    ref = ctypes.CDLL(so)
    ref.ref_nms.restype = ctypes.c_int
fp = lambda x: x.ctypes.data_as(ctypes.c_void_p)
for n in (1024, 4096):
    bx = boxes(n, 0.9 * n ** 0.5, 7)
    g = torch.from_numpy(bx).cuda()
    scores = torch.linspace(1, 0, n).cuda()
    for _ in range(3):
        keep = mtt_amd.iou3d.nms_gpu(g, scores, 0.3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        keep = mtt_amd.iou3d.nms_gpu(g, scores, 0.3)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    line = f"nms_gpu n={n}: {dt * 1e6:9.1f} us / call ({len(keep)} kept, {n * (n - 1) / 2 / dt / 1e6:8.1f} M pairs/s)"
    if ref is not None:
        k = np.zeros(n, np.int64)
        t0 = time.perf_counter()
        kn = ref.ref_nms(n, fp(bx), ctypes.c_float(0.3), 1, fp(k))
        dc = time.perf_counter() - t0
        line += f"  |  reference device code on 1 host core: {dc * 1e6:11.1f} us ({kn} kept)  ->  x{dc / dt:.0f}"
        assert np.array_equal(keep.cpu().numpy(), k[:kn]), "kept sets differ"
    print(line, flush=True)
a, b = torch.from_numpy(boxes(2048, 40, 1)).cuda(), torch.from_numpy(boxes(2048, 40, 2)).cuda()
for _ in range(3):
    iou = mtt_amd.iou3d.boxes_iou_bev(a, b)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    iou = mtt_amd.iou3d.boxes_iou_bev(a, b)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
print(f"boxes_iou_bev 2048 x 2048: {dt * 1e6:9.1f} us / call ({2048 * 2048 / dt / 1e6:8.1f} M pairs/s)", flush=True)
