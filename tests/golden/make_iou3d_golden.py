"""Generate tests/golden/iou3d.npz from the REFERENCE's own device functions compiled for the host (oracle/build_ref_iou3d.sh; build
container only).  Seeded boxes [x1, y1, x2, y2, ry]: generic overlapping, nested, identical, edge-sharing, far apart, thin and degenerate
(zero-area) boxes; pairwise overlap / IoU matrices and the kept indices of both NMS variants at two thresholds."""
import ctypes
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def boxes(rng, n, spread=10.0):
    c = rng.uniform(-spread, spread, (n, 2))
    wh = rng.uniform(0.5, 6.0, (n, 2))
    ry = rng.uniform(-np.pi, np.pi, (n, 1))
    return np.concatenate([c - wh / 2, c + wh / 2, ry], 1).astype(np.float32)


def special():
    return np.array([[0, 0, 2, 2, 0.0], [0, 0, 2, 2, 0.0], [0, 0, 2, 2, np.pi / 2], [0.5, 0.5, 1.5, 1.5, 0.3], [2, 0, 4, 2, 0.0],
                     [1, 1, 3, 3, 0.0], [100, 100, 101, 101, 1.0], [0, 0, 2, 2, 1e-3], [0, 0, 8, 0.2, 0.7], [1, 1, 1, 1, 0.0],
                     [-1, -1, 3, 3, np.pi / 4], [0, 0, 2, 2, -3.0]], dtype=np.float32)


def main():
    so = subprocess.check_output(["bash", os.path.join(ROOT, "oracle", "build_ref_iou3d.sh")]).decode().strip().splitlines()[-1]
    lib = ctypes.CDLL(so)
    fp = lambda x: x.ctypes.data_as(ctypes.c_void_p)
    rng = np.random.default_rng(5)
    out = {}
    a, b = np.concatenate([special(), boxes(rng, 28, 4.0)]), np.concatenate([special()[::-1], boxes(rng, 21, 4.0)])
    ov, iou = np.zeros((len(a), len(b)), np.float32), np.zeros((len(a), len(b)), np.float32)
    lib.ref_boxes_overlap(len(a), fp(a), len(b), fp(b), fp(ov))
    lib.ref_boxes_iou_bev(len(a), fp(a), len(b), fp(b), fp(iou))
    out.update(pair_a=a, pair_b=b, overlap=ov, iou=iou)
    lib.ref_nms.restype = ctypes.c_int
    for tag, n, spread in (("small", 70, 6.0), ("large", 700, 25.0)):
        bx = boxes(rng, n, spread)
        out[f"nms_{tag}_boxes"] = bx
        for thr in (0.1, 0.5):
            for rot in (1, 0):
                keep = np.zeros(n, np.int64)
                k = lib.ref_nms(n, fp(bx), ctypes.c_float(thr), rot, fp(keep))
                out[f"nms_{tag}_t{thr}_r{rot}"] = keep[:k].copy()
    # 4 096 boxes at the density of tools/iou3d_bench.py: only the seed and the kept index sets are stored (the boxes regenerate from the
    # seed with `boxes`), rotated and axis-aligned, threshold 0.3
    n, seed = 4096, 7
    bx = boxes(np.random.default_rng(seed), n, 0.9 * n ** 0.5)
    for rot in (1, 0):
        keep = np.zeros(n, np.int64)
        k = lib.ref_nms(n, fp(bx), ctypes.c_float(0.3), rot, fp(keep))
        out[f"nms_4096_t0.3_r{rot}"] = keep[:k].astype(np.int32)
    out["nms_4096_seed"] = np.array([seed, n], np.int64)
    np.savez_compressed(os.path.join(HERE, "iou3d.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    sys.exit(main())
