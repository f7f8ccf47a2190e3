"""Host-side mirror of the reference's rotated-box ops (TaskPrompter/detection_toolbox/iou3d/iou3d_utils.py:7-72: same function names,
arguments and results) on the HIP kernels of csrc/iou3d.hip.  Boxes are [x1, y1, x2, y2, ry] fp32 on a HIP device.

Unlike the reference (which copies the N x N/64 suppression masks to the host and reduces them there), the greedy pass runs on the
device; the only host synchronisation is reading the number of kept boxes to size the returned index tensor."""
import ctypes

import torch

from . import _lib, ops


def _pairwise(boxes_a, boxes_b, iou):
    a, b = boxes_a.contiguous().float(), boxes_b.contiguous().float()
    out = a.new_zeros((a.shape[0], b.shape[0]))
    if a.shape[0] and b.shape[0]:
        ops.call("boxes_overlap_bev", args=[a, a.shape[0], b, b.shape[0], out, iou])
    return out


def boxes_overlap_bev(boxes_a, boxes_b):
    """Overlap areas (M, N) of rotated boxes (iou3d.cpp:52-72)."""
    return _pairwise(boxes_a, boxes_b, 0)


def boxes_iou_bev(boxes_a, boxes_b):
    """iou3d_utils.py:7-23."""
    return _pairwise(boxes_a, boxes_b, 1)


def nms_ws_bytes(n):
    lib = _lib.load()
    lib.mtt_nms_ws_bytes.restype = ctypes.c_size_t
    return int(lib.mtt_nms_ws_bytes(int(n)))


def _nms_sorted(boxes, thresh, rotated):
    """boxes [n, 5] sorted by descending score -> kept indices (int64, on the boxes' device)."""
    n = boxes.shape[0]
    if n == 0:
        return torch.zeros(0, dtype=torch.long, device=boxes.device)
    keep = torch.zeros(n, dtype=torch.long, device=boxes.device)
    num = torch.zeros(1, dtype=torch.int32, device=boxes.device)
    n_cb = (n + 63) // 64
    ws = torch.empty(n * n_cb, dtype=torch.int64, device=boxes.device)          # n * ceil(n/64) 64-bit mask words
    ops.call("nms_bev", args=[boxes, n, float(thresh), 1 if rotated else 0, keep, num, ws])
    return keep[:int(num.item())]


def nms_gpu(boxes, scores, thresh, pre_maxsize=None, post_max_size=None):
    """Rotated NMS (iou3d_utils.py:26-51): indices of the kept boxes, highest score first."""
    order = scores.sort(0, descending=True)[1]
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    keep = order[_nms_sorted(boxes[order].contiguous().float(), thresh, True)].contiguous()
    if post_max_size is not None:
        keep = keep[:post_max_size]
    return keep


def nms_normal_gpu(boxes, scores, thresh):
    """Axis-aligned NMS on the same box format (iou3d_utils.py:54-72)."""
    order = scores.sort(0, descending=True)[1]
    return order[_nms_sorted(boxes[order].contiguous().float(), thresh, False)].contiguous()
