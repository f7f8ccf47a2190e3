// Bird's-eye-view rotated-box overlap / IoU and the two NMS variants of the 3-D detection branch (replaces
// TaskPrompter/detection_toolbox/iou3d/src/iou3d_kernel.cu + the host reduction of iou3d.cpp).  Box = [x1, y1, x2, y2, ry].
// The geometric algorithm (edge crossings + contained corners, angular sort about the vertex mean, fan area; margins 1e-5 / 1e-8) is the
// reference's, so degenerate configurations (identical, nested, edge-sharing boxes) resolve the same way; what is different:
//   * pairwise kernels: one lane per (a, b) pair, 64 consecutive b per wave (coalesced result rows);
//   * NMS: a 64 x 64 tile of the suppression matrix per WAVE (64-bit masks = one word per lane), only the tiles on or above the diagonal
//     (the greedy pass never reads the others), the column boxes broadcast from LDS;
//   * the greedy pass runs ON THE DEVICE in one wave (bit test on an LDS word per box, the suppressor's mask row OR-ed in by all lanes):
//     keep indices and their count stay in device memory — the reference copies the N x N/64 mask to the host and synchronises.
#include "mtt_device.h"

namespace {

#pragma clang fp contract(off)     // same rounding sequence as the (non-fused) restatement the goldens were generated with

constexpr float EPS_ = 1e-8f, MARGIN_ = 1e-5f;
struct P2 { float x, y; };

MTT_DEV float cr3(P2 p1, P2 p2, P2 p0) { return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y); }

MTT_DEV void corners(const float* box, P2 (&c)[5]) {
  const float cx = (box[0] + box[2]) / 2, cy = (box[1] + box[3]) / 2;
  const float cs = cosf(box[4]), sn = sinf(box[4]);
  const float xs[4] = {box[0], box[2], box[2], box[0]}, ys[4] = {box[1], box[1], box[3], box[3]};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float dx = xs[k] - cx, dy = ys[k] - cy;
    c[k].x = dx * cs + dy * sn + cx;
    c[k].y = -dx * sn + dy * cs + cy;
  }
  c[4] = c[0];
}

MTT_DEV bool in_box(const float* box, P2 p) {
  const float cx = (box[0] + box[2]) / 2, cy = (box[1] + box[3]) / 2;
  const float cs = cosf(-box[4]), sn = sinf(-box[4]);
  const float dx = p.x - cx, dy = p.y - cy;
  const float rx = dx * cs + dy * sn + cx, ry = -dx * sn + dy * cs + cy;
  return rx > box[0] - MARGIN_ && rx < box[2] + MARGIN_ && ry > box[1] - MARGIN_ && ry < box[3] + MARGIN_;
}

MTT_DEV bool seg_x(P2 p1, P2 p0, P2 q1, P2 q0, P2& ans) {
  if (!(fminf(p0.x, p1.x) <= fmaxf(q0.x, q1.x) && fminf(q0.x, q1.x) <= fmaxf(p0.x, p1.x) &&
        fminf(p0.y, p1.y) <= fmaxf(q0.y, q1.y) && fminf(q0.y, q1.y) <= fmaxf(p0.y, p1.y))) return false;
  const float s1 = cr3(q0, p1, p0), s2 = cr3(p1, q1, p0), s3 = cr3(p0, q1, q0), s4 = cr3(q1, p1, q0);
  if (!(s1 * s2 > 0 && s3 * s4 > 0)) return false;
  const float s5 = cr3(q1, p1, p0);
  if (fabsf(s5 - s1) > EPS_) {
    ans.x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
    ans.y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
  } else {
    const float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
    const float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
    const float D = a0 * b1 - a1 * b0;
    ans.x = (b0 * c1 - b1 * c0) / D;
    ans.y = (a1 * c0 - a0 * c1) / D;
  }
  return true;
}

MTT_DEV float overlap_area(const float* A, const float* B) {
  P2 ca[5], cb[5], pts[16];
  corners(A, ca);
  corners(B, cb);
  int n = 0;
  float sx = 0.f, sy = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      P2 ip;
      if (seg_x(ca[i + 1], ca[i], cb[j + 1], cb[j], ip)) { pts[n++] = ip; sx += ip.x; sy += ip.y; }
    }
  for (int k = 0; k < 4; ++k) {
    if (in_box(A, cb[k])) { pts[n++] = cb[k]; sx += cb[k].x; sy += cb[k].y; }
    if (in_box(B, ca[k])) { pts[n++] = ca[k]; sx += ca[k].x; sy += ca[k].y; }
  }
  if (n == 0) return 0.f;
  const float mx = sx / n, my = sy / n;
  float ang[16];
  for (int k = 0; k < n; ++k) ang[k] = atan2f(pts[k].y - my, pts[k].x - mx);
  for (int j = 0; j < n - 1; ++j)                    // the reference's bubble sort (strict >): the order of equal angles is part of the result
    for (int i = 0; i < n - j - 1; ++i)
      if (ang[i] > ang[i + 1]) {
        const P2 tp = pts[i]; pts[i] = pts[i + 1]; pts[i + 1] = tp;
        const float ta = ang[i]; ang[i] = ang[i + 1]; ang[i + 1] = ta;
      }
  float area = 0.f;
  for (int k = 0; k < n - 1; ++k)
    area += (pts[k].x - pts[0].x) * (pts[k + 1].y - pts[0].y) - (pts[k].y - pts[0].y) * (pts[k + 1].x - pts[0].x);
  return fabsf(area) / 2.0f;
}

MTT_DEV float iou_rot(const float* A, const float* B) {
  const float sa = (A[2] - A[0]) * (A[3] - A[1]), sb = (B[2] - B[0]) * (B[3] - B[1]);
  const float ov = overlap_area(A, B);
  return ov / fmaxf(sa + sb - ov, EPS_);
}
MTT_DEV float iou_axis(const float* a, const float* b) {
  const float w = fmaxf(fminf(a[2], b[2]) - fmaxf(a[0], b[0]), 0.f), h = fmaxf(fminf(a[3], b[3]) - fmaxf(a[1], b[1]), 0.f);
  const float inter = w * h;
  const float sa = (a[2] - a[0]) * (a[3] - a[1]), sb = (b[2] - b[0]) * (b[3] - b[1]);
  return inter / fmaxf(sa + sb - inter, EPS_);
}

__global__ __launch_bounds__(256) void pairwise_kernel(const float* a, int na, const float* b, int nb, float* out, int iou) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= (int64_t)na * nb) return;
  const int i = (int)(t / nb), j = (int)(t % nb);
  float A[5], B[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) { A[k] = a[i * 5 + k]; B[k] = b[j * 5 + k]; }
  out[t] = iou ? iou_rot(A, B) : overlap_area(A, B);
}

// one wave per 64 x 64 tile (rb <= cb): mask[(rb*64 + lane) * col_blocks + cb] bit j set when box rb*64+lane suppresses box cb*64+j
__global__ __launch_bounds__(64) void nms_mask_kernel(const float* boxes, int n, float thresh, int rotated, unsigned long long* mask) {
  const int cb = blockIdx.x, rb = blockIdx.y;
  if (rb > cb) return;
  const int col_blocks = (n + 63) / 64;
  __shared__ float cols[64 * 5];
  const int lane = threadIdx.x;
  const int cn = min(n - cb * 64, 64), rn = min(n - rb * 64, 64);
  if (lane < cn)
#pragma unroll
    for (int k = 0; k < 5; ++k) cols[lane * 5 + k] = boxes[(int64_t)(cb * 64 + lane) * 5 + k];
  __syncthreads();
  if (lane >= rn) return;
  float A[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) A[k] = boxes[(int64_t)(rb * 64 + lane) * 5 + k];
  unsigned long long m = 0;
  for (int j = (rb == cb ? lane + 1 : 0); j < cn; ++j) {
    const float v = rotated ? iou_rot(A, cols + j * 5) : iou_axis(A, cols + j * 5);
    if (v > thresh) m |= 1ull << j;
  }
  mask[(int64_t)(rb * 64 + lane) * col_blocks + cb] = m;
}

// greedy pass, one wave: remv (one 64-bit word per column block) in LDS; lanes OR the kept box's mask row in parallel
__global__ __launch_bounds__(64) void nms_reduce_kernel(const unsigned long long* mask, int n, long long* keep, int* num_out) {
  extern __shared__ unsigned long long remv[];
  const int col_blocks = (n + 63) / 64;
  const int lane = threadIdx.x;
  for (int j = lane; j < col_blocks; j += 64) remv[j] = 0ull;
  __syncthreads();
  int k = 0;
  for (int i = 0; i < n; ++i) {
    const int nb = i >> 6;
    const bool dead = (remv[nb] >> (i & 63)) & 1ull;      // wave-uniform
    if (!dead) {
      if (lane == 0) keep[k] = i;
      ++k;
      for (int j = nb + lane; j < col_blocks; j += 64) remv[j] |= mask[(int64_t)i * col_blocks + j];
    }
    __syncthreads();
  }
  if (lane == 0) *num_out = k;
}

}  // namespace

extern "C" int mtt_boxes_overlap_bev(const float* a, int na, const float* b, int nb, float* out, int iou, void* stream) {
  if (!a || !b || !out || na <= 0 || nb <= 0) return MTT_E_BADARG;
  const int64_t n = (int64_t)na * nb;
  hipLaunchKernelGGL(pairwise_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, na, b, nb, out, iou);
  return (int)hipGetLastError();
}

extern "C" size_t mtt_nms_ws_bytes(int n) { return n <= 0 ? 0 : (size_t)n * ((n + 63) / 64) * 8; }

extern "C" int mtt_nms_bev(const float* boxes, int n, float thresh, int rotated, long long* keep, int* num_out, void* ws, void* stream) {
  if (!boxes || !keep || !num_out || !ws || n <= 0) return MTT_E_BADARG;
  const int cb = (n + 63) / 64;
  if ((size_t)cb * 8 > 64 * 1024) return MTT_E_UNSUPPORTED;        /* > 524 288 boxes */
  hipLaunchKernelGGL(nms_mask_kernel, dim3(cb, cb), dim3(64), 0, (hipStream_t)stream, boxes, n, thresh, rotated, (unsigned long long*)ws);
  hipLaunchKernelGGL(nms_reduce_kernel, dim3(1), dim3(64), cb * 8, (hipStream_t)stream, (const unsigned long long*)ws, n, keep, num_out);
  return (int)hipGetLastError();
}
