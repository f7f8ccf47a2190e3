// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of libmtt_hip.so.
//
// MFMA conventions used everywhere (v_mfma_f32_16x16x32_bf16, cdna_hip_programming.md §3):
//   lane l:  i = l & 15, g = l >> 4
//   A fragment (16 x 32): lane holds A[i][8g .. 8g+7]      (8 bf16, one 16-byte LDS read)
//   B fragment (32 x 16): lane holds B[8g .. 8g+7][i]
//   C/D (16 x 16)       : acc[r] = D[4g + r][i]
// Both operands are staged into LDS as row-major [row][64 k] bf16 tiles (128 B per row) whose
// 16-byte chunks are XOR-swizzled so that the 16-lane groups of ds_read_b128 hit 16 distinct slots.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include "mtt_hip.h"

// Host: opt a kernel in to more than 64 KiB of dynamic LDS, once per device (the attribute is per device, and a process may
// drive several).  `done` is a per-call-site bit mask of devices already configured: a cache of an idempotent setup call, safe
// under concurrent first use (two threads may both set the attribute; the result is the same).
static inline int mtt_ensure_dyn_lds(const void* fn, int bytes, std::atomic<unsigned long long>& done) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return (int)hipGetLastError();
  const unsigned long long bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return 0;
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return (int)e;
  done.fetch_or(bit, std::memory_order_release);
  return 0;
}

typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define MTT_DEV __device__ __forceinline__

MTT_DEV float bf2f(bf16_t h) { return __builtin_bit_cast(float, ((unsigned)h) << 16); }
MTT_DEV bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
// two floats -> packed bf16 pair (one v_cvt_pk_bf16_f32)
MTT_DEV unsigned pack2(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){a, b}, bf16x2_t));
}
MTT_DEV float lo_of(unsigned u) { return __builtin_bit_cast(float, u << 16); }
MTT_DEV float hi_of(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

// load/store one element of runtime dtype
MTT_DEV float ld_elem(const void* p, int64_t idx, int dtype) {
  return dtype == MTT_F32 ? ((const float*)p)[idx] : bf2f(((const bf16_t*)p)[idx]);
}
MTT_DEV void st_elem(void* p, int64_t idx, int dtype, float v) {
  if (dtype == MTT_F32) ((float*)p)[idx] = v; else ((bf16_t*)p)[idx] = f2bf(v);
}

// erf via the rational approximation x*P(x^2)/Q(x^2) on [-4, 4] (|error| < 4.5e-7, checked against scipy in
// tools/check_fast_erf.py): 12 FMAs + one v_rcp_f32 instead of libm's branchy erff — the GELU / GELU' epilogues and the
// BatchNorm+GELU row kernels were ALU-bound on erff.
MTT_DEV float fast_erf(float x) {
  x = fminf(fmaxf(x, -4.0f), 4.0f);
  const float x2 = x * x;
  float p = fmaf(x2, -2.72614225801306e-10f, 2.77068142495902e-08f);
  p = fmaf(x2, p, -2.10102402082508e-06f);
  p = fmaf(x2, p, -5.69250639462346e-05f);
  p = fmaf(x2, p, -7.34990630326855e-04f);
  p = fmaf(x2, p, -2.95459980854025e-03f);
  p = fmaf(x2, p, -1.60960333262415e-02f);
  float q = fmaf(x2, -1.45660718464996e-05f, -2.13374055278905e-04f);
  q = fmaf(x2, q, -1.68282697438203e-03f);
  q = fmaf(x2, q, -7.37332916720468e-03f);
  q = fmaf(x2, q, -1.42647390514189e-02f);
  return x * p * __builtin_amdgcn_rcpf(q);
}
MTT_DEV float gelu_f(float x) { return 0.5f * x * (1.0f + fast_erf(x * 0.70710678118654752f)); }
// GELU(x) and GELU'(x) together (one erf): the fc1 epilogue of MTT_ACT_GELU_DAUX
MTT_DEV void gelu_both_f(float x, float& g, float& dg) {
  const float c = 0.5f * (1.0f + fast_erf(x * 0.70710678118654752f));
  g = x * c;
  dg = fmaf(x * 0.3989422804014327f, __expf(-0.5f * x * x), c);
}
MTT_DEV float gelu_grad_f(float x) {
  return 0.5f * (1.0f + fast_erf(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

// ---- LDS tile addressing: [rows][64] bf16, 8 chunks of 16 B per row, swizzled -------------------
MTT_DEV int lds_swz(int row) { return ((row >> 1) ^ (row >> 4)) & 7; }
MTT_DEV int lds_off(int row, int chunk) { return row * 128 + (((chunk ^ lds_swz(row)) & 7) << 4); }

// ---- exact unsigned division by a runtime constant (magic computed on the host) -----------------
struct FastDiv { uint32_t magic, shift, d; };
MTT_DEV uint32_t fdiv(uint32_t n, FastDiv f) { return (__umulhi(n, f.magic) + n) >> f.shift; }

// ---- 8 consecutive elements (runtime dtype) -> packed bf16 hi (and lo = residual for X3) ---------
template <bool X3>
MTT_DEV void load8(const void* base, int64_t idx, int dtype, bool ok, u32x4& hi, u32x4& lo) {
  hi = (u32x4){0u, 0u, 0u, 0u};
  if (X3) lo = hi;
  if (!ok) return;
  if (dtype == MTT_BF16) {
    hi = *(const u32x4*)((const bf16_t*)base + idx);
  } else {
    const float4 v0 = *(const float4*)((const float*)base + idx);
    const float4 v1 = *(const float4*)((const float*)base + idx + 4);
    hi = (u32x4){pack2(v0.x, v0.y), pack2(v0.z, v0.w), pack2(v1.x, v1.y), pack2(v1.z, v1.w)};
    if (X3) {
      lo = (u32x4){pack2(v0.x - lo_of(hi.x), v0.y - hi_of(hi.x)), pack2(v0.z - lo_of(hi.y), v0.w - hi_of(hi.y)),
                   pack2(v1.x - lo_of(hi.z), v1.y - hi_of(hi.z)), pack2(v1.z - lo_of(hi.w), v1.w - hi_of(hi.w))};
    }
  }
}

// Raw 8-element fetch for the register stagers: branch-free (invalid chunks read element 0 of the tensor
// and are zeroed at conversion time), conversion / X3 split deferred so that all loads of a K step are
// in flight together.  The source dtype is a compile-time parameter (bf16: 16 B, f32: 32 B per chunk).
template <bool F32> struct Raw8;
template <> struct Raw8<false> { u32x4 r0; };
template <> struct Raw8<true> { float4 v0, v1; };

template <bool F32>
MTT_DEV void load8_raw(const void* base, int64_t idx, bool ok, Raw8<F32>& r) {
  idx = ok ? idx : 0;
  if constexpr (F32) {
    r.v0 = *(const float4*)((const float*)base + idx);
    r.v1 = *(const float4*)((const float*)base + idx + 4);
  } else {
    r.r0 = *(const u32x4*)((const bf16_t*)base + idx);
  }
}
template <bool X3, bool F32>
MTT_DEV void cvt8(bool ok, const Raw8<F32>& r, u32x4& hi, u32x4& lo) {
  const u32x4 zero = (u32x4){0u, 0u, 0u, 0u};
  if (X3) lo = zero;
  if constexpr (!F32) {
    hi = ok ? r.r0 : zero;
  } else {
    const float f0 = r.v0.x, f1 = r.v0.y, f2 = r.v0.z, f3 = r.v0.w, f4 = r.v1.x, f5 = r.v1.y, f6 = r.v1.z, f7 = r.v1.w;
    hi = (u32x4){pack2(f0, f1), pack2(f2, f3), pack2(f4, f5), pack2(f6, f7)};
    if (X3) {
      lo = (u32x4){pack2(f0 - lo_of(hi.x), f1 - hi_of(hi.x)), pack2(f2 - lo_of(hi.y), f3 - hi_of(hi.y)),
                   pack2(f4 - lo_of(hi.z), f5 - hi_of(hi.z)), pack2(f6 - lo_of(hi.w), f7 - hi_of(hi.w))};
      lo = ok ? lo : zero;
    }
    hi = ok ? hi : zero;
  }
}

// 4 (k) x 8 (row) block held as 4 packed rows -> 8 pieces of 4 k-consecutive bf16 (one per row)
MTT_DEV void transpose4x8(const u32x4 (&in)[4], u32x2 (&out)[8]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const unsigned a0 = in[0][q], a1 = in[1][q], a2 = in[2][q], a3 = in[3][q];
    out[2 * q][0] = (a0 & 0xffffu) | (a1 << 16);
    out[2 * q][1] = (a2 & 0xffffu) | (a3 << 16);
    out[2 * q + 1][0] = (a0 >> 16) | (a1 & 0xffff0000u);
    out[2 * q + 1][1] = (a2 >> 16) | (a3 & 0xffff0000u);
  }
}

// ---- all-reduce steps across the 4 lane groups (lanes l, l ^ 16, l ^ 32, l ^ 48) without the LDS pipe (__shfl_xor compiles to ds_bpermute_b32 + a full lgkmcnt wait):
// v_permlane16_swap exchanges the odd 16-lane rows of its first operand with the even rows of the second, v_permlane32_swap the upper
// 32 lanes of the first with the lower 32 of the second (profiles/r02_probe_permlane_swap.txt), so with both operands = x the two results
// hold x and its xor-16 (xor-32) partner.  Inline asm: hipcc folds repeated calls of the builtins.
MTT_DEV void xor16_pair(float x, float& a, float& b) {
  a = x; b = x;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
MTT_DEV void xor32_pair(float x, float& a, float& b) {
  a = x; b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
MTT_DEV float groups_max(float x) {
  float a, b;
  xor16_pair(x, a, b); x = fmaxf(a, b);
  xor32_pair(x, a, b); return fmaxf(a, b);
}
MTT_DEV float groups_sum(float x) {
  float a, b;
  xor16_pair(x, a, b); x = a + b;
  xor32_pair(x, a, b); return a + b;
}

MTT_DEV f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// XCD-aware bijective remap of a 1-D block id (8 XCDs, block b runs on XCD b % 8)
MTT_DEV int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, in = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + in;
}

// Grouped tile order (GROUP_M rows of tiles are swept together): the ~32 workgroups an XCD runs concurrently then cover a
// GROUP_M x (32 / GROUP_M) patch of output tiles and share both operands' panels in that XCD's L2, instead of one row of
// tiles that re-streams all of B per row (PMC: 5x algorithmic FETCH_SIZE with the row-major order).
MTT_DEV void grouped_tile(int wg, int tiles_m, int tiles_n, int gm, int& tile_m, int& tile_n) {
  const int per_group = gm * tiles_n;
  const int group = wg / per_group;
  const int first_m = group * gm;
  const int rows = tiles_m - first_m < gm ? tiles_m - first_m : gm;
  const int in = wg - group * per_group;
  tile_m = first_m + in % rows;
  tile_n = in / rows;
}

MTT_DEV float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
MTT_DEV float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- deterministic second stage of a cross-workgroup reduction (replaces fp32 atomics: the sums below are taken in a FIXED order, so a
// kernel's result does not depend on the order in which its workgroups happen to finish).  First stage: workgroup s writes its partial
// results as plane s of a caller-owned workspace, ws[s * n + i].
//   few : one thread per output i, looping over the S planes (S small, n large)
//   many: one workgroup per output i, S partials summed with a fixed 256-way tree (n small, S large)
static __global__ __launch_bounds__(256) void mtt_reduce_few_kernel(const float* ws, int S, int64_t n, float* dst, int accumulate) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float s = 0.f;
    for (int k = 0; k < S; ++k) s += ws[(int64_t)k * n + i];
    dst[i] = accumulate ? dst[i] + s : s;
  }
}
static __global__ __launch_bounds__(256) void mtt_reduce_many_kernel(const float* ws, int S, int n, float* dst, float scale, int accumulate) {
  __shared__ float red[256];
  const int i = blockIdx.x;
  float s = 0.f;
  for (int k = threadIdx.x; k < S; k += 256) s += ws[(int64_t)k * n + i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) dst[i] = accumulate ? dst[i] + scale * red[0] : scale * red[0];
}
// column sums, second stage: ws [nblk][pad8(cols)] per-row-block partials -> dst[cols]; 32 columns per workgroup, 8 row groups summed in order
static __global__ __launch_bounds__(256) void mtt_colsum_final_kernel(const float* ws, float* dst, int cols, int nblk, int64_t dst_zs) {
  __shared__ float sh[8][32];
  ws += (int64_t)blockIdx.z * nblk * (((int64_t)cols + 7) / 8 * 8);
  dst += (int64_t)blockIdx.z * dst_zs;
  const int cl = threadIdx.x & 31, pl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  const int64_t colsP = ((int64_t)cols + 7) / 8 * 8;
  float t = 0.f;
  if (c < cols) {                                       // 4 independent accumulators: 4 loads in flight per lane, fixed order
    float u[4] = {0.f, 0.f, 0.f, 0.f};
    int b = pl;
    for (; b + 24 < nblk; b += 32) {
#pragma unroll
      for (int k = 0; k < 4; ++k) u[k] += ws[(int64_t)(b + 8 * k) * colsP + c];
    }
    for (; b < nblk; b += 8) u[0] += ws[(int64_t)b * colsP + c];
    t = (u[0] + u[1]) + (u[2] + u[3]);
  }
  sh[pl][cl] = t;
  __syncthreads();
  if (pl == 0 && c < cols) {
#pragma unroll
    for (int l = 1; l < 8; ++l) t += sh[l][cl];
    dst[c] = t;
  }
}
static inline unsigned mtt_reduce_few_grid(int64_t n) { const int64_t g = (n + 255) / 256; return (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g)); }
