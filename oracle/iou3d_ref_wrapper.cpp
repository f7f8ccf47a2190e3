// TEST INFRASTRUCTURE ONLY — host build of the REFERENCE's own device functions (box_overlap, iou_bev, iou_normal of
// TaskPrompter/detection_toolbox/iou3d/src/iou3d_kernel.cu:18-239,323-331) so that the HIP kernels and the numpy restatement can be
// checked against the unmodified reference arithmetic.  oracle/build_ref_iou3d.sh cuts the device-function part of the .cu (everything
// before the first __global__ kernel) into oracle/_ref/iou3d_device.inc — a build output, never committed — and compiles this file
// with g++; the CUDA qualifiers are defined away, the kernels' per-thread bodies are replayed by plain loops below.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>
#define __device__
#define __global__
#define __shared__
using std::max;
using std::min;
#include "_ref/iou3d_device.inc"

extern "C" {
// boxes_overlap_kernel / boxes_iou_bev_kernel (iou3d_kernel.cu:241-275): ans[a * num_b + b]
void ref_boxes_overlap(int num_a, const float* a, int num_b, const float* b, float* ans) {
  for (int i = 0; i < num_a; ++i)
    for (int j = 0; j < num_b; ++j) ans[(int64_t)i * num_b + j] = box_overlap(a + i * 5, b + j * 5);
}
void ref_boxes_iou_bev(int num_a, const float* a, int num_b, const float* b, float* ans) {
  for (int i = 0; i < num_a; ++i)
    for (int j = 0; j < num_b; ++j) ans[(int64_t)i * num_b + j] = iou_bev(a + i * 5, b + j * 5);
}
// nms_kernel + the host reduction of iou3d.cpp:103-152 (rotated = 1) / nms_normal_kernel + :154-203 (rotated = 0); boxes sorted by score
int ref_nms(int n, const float* boxes, float thresh, int rotated, int64_t* keep) {
  const int cb = (n + 63) / 64;
  std::vector<unsigned long long> mask((size_t)n * cb, 0ull), remv(cb, 0ull);
  for (int i = 0; i < n; ++i)
    for (int j = i + 1; j < n; ++j) {
      const float v = rotated ? iou_bev(boxes + i * 5, boxes + j * 5) : iou_normal(boxes + i * 5, boxes + j * 5);
      if (v > thresh) mask[(size_t)i * cb + j / 64] |= 1ull << (j % 64);
    }
  int k = 0;
  for (int i = 0; i < n; ++i) {
    if (!(remv[i / 64] & (1ull << (i % 64)))) {
      keep[k++] = i;
      for (int j = i / 64; j < cb; ++j) remv[j] |= mask[(size_t)i * cb + j];
    }
  }
  return k;
}
}
