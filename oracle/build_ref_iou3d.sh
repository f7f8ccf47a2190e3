#!/bin/bash
# TEST INFRASTRUCTURE ONLY — compile the reference's iou3d device functions for the host (build container only; the GPU box has no
# /root/reference and uses the committed fixtures tests/golden/iou3d.npz).  Outputs go to oracle/_ref/ (git-ignored).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
REF="${MTT_REFERENCE_ROOT:-/root/reference}/TaskPrompter/detection_toolbox/iou3d/src/iou3d_kernel.cu"
[ -f "$REF" ] || { echo "reference not found: $REF" >&2; exit 3; }
mkdir -p "$HERE/_ref"
# device functions of the .cu up to the first kernel; then iou_normal (between the two NMS kernels)
L1=$(grep -n '^__global__ void boxes_overlap_kernel' "$REF" | cut -d: -f1)
A=$(grep -n '^__device__ inline float iou_normal' "$REF" | cut -d: -f1)
B=$(grep -n '^__global__ void nms_normal_kernel' "$REF" | cut -d: -f1)
{ head -n $((L1 - 1)) "$REF"; sed -n "${A},$((B - 1))p" "$REF"; } > "$HERE/_ref/iou3d_device.inc"
g++ -O2 -fPIC -shared -ffp-contract=off -I"$HERE" "$HERE/iou3d_ref_wrapper.cpp" -o "$HERE/_ref/libiou3d_ref.so"
echo "$HERE/_ref/libiou3d_ref.so"
