"""TEST INFRASTRUCTURE ONLY (CPU oracle) — restatement of the reference's bird's-eye-view rotated-box overlap / IoU and the two NMS
variants (TaskPrompter/detection_toolbox/iou3d/src/iou3d_kernel.cu:18-239 device functions, :241-275 pairwise kernels, :277-321 /
:333-377 NMS mask kernels; iou3d.cpp:103-203 greedy reduction; iou3d_utils.py:7-72 Python wrappers).

Plain float32 arithmetic in pure-Python loops: SMALL cases only (a few hundred pairs).  PINNED: oracle/build_ref_iou3d.sh compiles the
reference's own device functions for the host (g++ on the .cu's device-function part, qualifiers defined away);
tests/golden/make_iou3d_golden.py runs that build on seeded boxes (generic, nested, identical, edge-sharing, far apart, degenerate)
and commits inputs + outputs as tests/golden/iou3d.npz; tests/test_iou3d.py checks this restatement and (on the GPU box) the HIP
kernels against them.  A box is [x1, y1, x2, y2, ry]: the axis-aligned rectangle (x1, y1)-(x2, y2) rotated by ry about its centre.
"""
import math

import numpy as np

f32 = np.float32
EPS = f32(1e-8)          # iou3d_kernel.cu:17
MARGIN = f32(1e-5)       # :52


def _cross2(ax, ay, bx, by):
    return f32(f32(ax * by) - f32(ay * bx))


def _cross3(p1, p2, p0):          # (p1 - p0) x (p2 - p0)   (:38-41)
    return f32(f32(f32(p1[0] - p0[0]) * f32(p2[1] - p0[1])) - f32(f32(p2[0] - p0[0]) * f32(p1[1] - p0[1])))


def _corners(box):
    """:141-168: corners rotated about the centre with (cos ry, sin ry): x' = dx cos + dy sin + cx, y' = -dx sin + dy cos + cy."""
    x1, y1, x2, y2, ang = (f32(v) for v in box)
    cx, cy = f32(f32(x1 + x2) / f32(2)), f32(f32(y1 + y2) / f32(2))
    c, s = f32(math.cos(float(ang))), f32(math.sin(float(ang)))
    out = []
    for (px, py) in ((x1, y1), (x2, y1), (x2, y2), (x1, y2)):
        dx, dy = f32(px - cx), f32(py - cy)
        out.append((f32(f32(f32(dx * c) + f32(dy * s)) + cx), f32(f32(f32(-dx * s) + f32(dy * c)) + cy)))
    return out


def _in_box(box, p):
    """:50-73: rotate the point back by -(-ry) ... the reference rotates with cos(-ry), sin(-ry) and a 1e-5 margin."""
    x1, y1, x2, y2, ang = (f32(v) for v in box)
    cx, cy = f32(f32(x1 + x2) / f32(2)), f32(f32(y1 + y2) / f32(2))
    c, s = f32(math.cos(float(-ang))), f32(math.sin(float(-ang)))
    dx, dy = f32(p[0] - cx), f32(p[1] - cy)
    rx = f32(f32(f32(dx * c) + f32(dy * s)) + cx)
    ry = f32(f32(f32(-dx * s) + f32(dy * c)) + cy)
    return rx > f32(x1 - MARGIN) and rx < f32(x2 + MARGIN) and ry > f32(y1 - MARGIN) and ry < f32(y2 + MARGIN)


def _segment_intersection(p1, p0, q1, q0):
    """:75-106: strict crossing test on the four signed areas, then the intersection point."""
    if not (min(p0[0], p1[0]) <= max(q0[0], q1[0]) and min(q0[0], q1[0]) <= max(p0[0], p1[0]) and
            min(p0[1], p1[1]) <= max(q0[1], q1[1]) and min(q0[1], q1[1]) <= max(p0[1], p1[1])):
        return None
    s1, s2 = _cross3(q0, p1, p0), _cross3(p1, q1, p0)
    s3, s4 = _cross3(p0, q1, q0), _cross3(q1, p1, q0)
    if not (f32(s1 * s2) > 0 and f32(s3 * s4) > 0):
        return None
    s5 = _cross3(q1, p1, p0)
    if abs(f32(s5 - s1)) > EPS:
        d = f32(s5 - s1)
        return (f32(f32(f32(s5 * q0[0]) - f32(s1 * q1[0])) / d), f32(f32(f32(s5 * q0[1]) - f32(s1 * q1[1])) / d))
    a0, b0, c0 = f32(p0[1] - p1[1]), f32(p1[0] - p0[0]), f32(f32(p0[0] * p1[1]) - f32(p1[0] * p0[1]))
    a1, b1, c1 = f32(q0[1] - q1[1]), f32(q1[0] - q0[0]), f32(f32(q0[0] * q1[1]) - f32(q1[0] * q0[1]))
    D = f32(f32(a0 * b1) - f32(a1 * b0))
    return (f32(f32(f32(b0 * c1) - f32(b1 * c0)) / D), f32(f32(f32(a1 * c0) - f32(a0 * c1)) / D))


def box_overlap(box_a, box_b):
    """:124-229: intersection polygon = edge crossings + corners of one box inside the other, sorted by angle about their mean
    (bubble sort with a strict 'greater' comparison, as the reference), area by the fan from the first vertex."""
    ca, cb = _corners(box_a), _corners(box_b)
    ca.append(ca[0])
    cb.append(cb[0])
    pts = []
    for i in range(4):
        for j in range(4):
            ip = _segment_intersection(ca[i + 1], ca[i], cb[j + 1], cb[j])
            if ip is not None:
                pts.append(ip)
    for k in range(4):
        if _in_box(box_a, cb[k]):
            pts.append(cb[k])
        if _in_box(box_b, ca[k]):
            pts.append(ca[k])
    n = len(pts)
    if n == 0:
        return f32(0)           # the reference divides 0 by 0 and sums an empty fan: |0| / 2
    sx = sy = f32(0)
    for (x, y) in pts:
        sx, sy = f32(sx + x), f32(sy + y)
    mx, my = f32(sx / f32(n)), f32(sy / f32(n))
    ang = lambda p: f32(math.atan2(float(f32(p[1] - my)), float(f32(p[0] - mx))))
    for j in range(n - 1):
        for i in range(n - j - 1):
            if ang(pts[i]) > ang(pts[i + 1]):
                pts[i], pts[i + 1] = pts[i + 1], pts[i]
    area = f32(0)
    for k in range(n - 1):
        area = f32(area + _cross2(f32(pts[k][0] - pts[0][0]), f32(pts[k][1] - pts[0][1]),
                                  f32(pts[k + 1][0] - pts[0][0]), f32(pts[k + 1][1] - pts[0][1])))
    return f32(abs(area) / f32(2))


def iou_bev(a, b):
    """:231-239."""
    sa = f32(f32(f32(a[2]) - f32(a[0])) * f32(f32(a[3]) - f32(a[1])))
    sb = f32(f32(f32(b[2]) - f32(b[0])) * f32(f32(b[3]) - f32(b[1])))
    ov = box_overlap(a, b)
    return f32(ov / max(f32(f32(sa + sb) - ov), EPS))


def iou_normal(a, b):
    """:323-331 (axis-aligned; the angle is ignored)."""
    a, b = [f32(v) for v in a], [f32(v) for v in b]
    w = max(f32(min(a[2], b[2]) - max(a[0], b[0])), f32(0))
    h = max(f32(min(a[3], b[3]) - max(a[1], b[1])), f32(0))
    inter = f32(w * h)
    sa, sb = f32(f32(a[2] - a[0]) * f32(a[3] - a[1])), f32(f32(b[2] - b[0]) * f32(b[3] - b[1]))
    return f32(inter / max(f32(f32(sa + sb) - inter), EPS))


def pairwise(boxes_a, boxes_b, iou):
    fn = iou_bev if iou else box_overlap
    return np.array([[fn(a, b) for b in boxes_b] for a in boxes_a], dtype=np.float32).reshape(len(boxes_a), len(boxes_b))


def nms(boxes, thresh, rotated=True):
    """nms_kernel / nms_normal_kernel + the greedy pass of iou3d.cpp:130-146 on boxes ALREADY sorted by descending score:
    box i suppresses every later box j with iou(i, j) > thresh unless i itself was suppressed.  Returns the kept indices."""
    fn = iou_bev if rotated else iou_normal
    n = len(boxes)
    removed = [False] * n
    keep = []
    for i in range(n):
        if removed[i]:
            continue
        keep.append(i)
        for j in range(i + 1, n):
            if not removed[j] and fn(boxes[i], boxes[j]) > f32(thresh):
                removed[j] = True
    return np.array(keep, dtype=np.int64)
