// ConvHead's first stage on the low-resolution task features, "taps first" (include/mtt_hip.h: mtt_upconv_desc):
//   conv3x3(up4(x)) = bias + sum_taps shift_tap(up4(W_tap x))
// The channel mixing W_tap x is a plain GEMM on the h x w map (mtt_gemm); the kernels here are the HBM-bound remainder: the sum of
// the nine shifted x4 bilinear expansions (forward) and its adjoint (backward).  Both exploit that the expansion is separable and that
// for an integer scale of 4 its weights are the constants 1/8, 3/8, 5/8, 7/8:
//
//   hi-res index 4j + r, r in [-1, 4], interpolates the two low-res "slots" (0: j-1, 1: j, 2: j+1)
//        r = -1: (0: 5/8, 1: 3/8)   r = 0: (0: 3/8, 1: 5/8)   r = 1: (0: 1/8, 1: 7/8)
//        r =  2: (1: 7/8, 2: 1/8)   r = 3: (1: 5/8, 2: 3/8)   r = 4: (1: 3/8, 2: 5/8)
//   PyTorch's align_corners=False clamps the source index at the borders, which is the same as replicating the border pixel of the
//   low-res map: slots are addressed with clamped indices and the weights stay constant.  The conv's zero padding acts on the hi-res
//   map: taps that fall on hi-res row / column -1 or 4h / 4w contribute nothing (masks on r = -1 at j = 0 and r = 4 at j = w - 1).
//
// One lane owns one low-res pixel x 4 channels: forward it produces the 4 x 4 hi-res pixels of that cell from the
// 3 x 3 low-res neighbourhood of all nine tap planes (column pass into 4 values per (tap row, low-res row), then row pass: 19.5 FMAs per
// output instead of 36); backward it gathers the nine tap gradients of its pixel from the 10 x 10 hi-res window around the cell.
// Workgroups are numbered so that all cells of one image run on one XCD (its L2 then serves the 3x / 6x re-reads of neighbouring rows).
#include "mtt_device.h"

namespace {

// table index r1 = r + 1 in [0, 5]
MTT_DEV constexpr int slot_a(int r1) { return r1 < 3 ? 0 : 1; }
MTT_DEV constexpr int slot_b(int r1) { return r1 < 3 ? 1 : 2; }
MTT_DEV constexpr float wgt_a(int r1) { return r1 == 0 ? 0.625f : r1 == 1 ? 0.375f : r1 == 2 ? 0.125f : r1 == 3 ? 0.875f : r1 == 4 ? 0.625f : 0.375f; }
MTT_DEV constexpr float wgt_b(int r1) { return 1.0f - wgt_a(r1); }

// CH = 4 channels per lane (8 B of bf16 / 16 B of fp32): 64 accumulators in the forward kernel, so 4 waves per SIMD stay resident
// and occupancy hides the L2 latency of the neighbourhood loads.  Addresses are a block-uniform 64-bit image base plus a 32-bit lane
// offset in elements (the host checks that one image's planes stay below 2^31 bytes).
constexpr int CH = 4;
template <bool BF> struct Raw;
template <> struct Raw<true> {
  u32x2 u;
  MTT_DEV void load(const void* base, unsigned off) { u = *(const u32x2*)((const bf16_t*)base + off); }
  MTT_DEV void zero() { u = (u32x2){0u, 0u}; }
  MTT_DEV void get(float (&v)[CH]) const { v[0] = lo_of(u[0]); v[1] = hi_of(u[0]); v[2] = lo_of(u[1]); v[3] = hi_of(u[1]); }
};
template <> struct Raw<false> {
  float4 a;
  MTT_DEV void load(const void* base, unsigned off) { a = *(const float4*)((const float*)base + off); }
  MTT_DEV void zero() { a = make_float4(0.f, 0.f, 0.f, 0.f); }
  MTT_DEV void get(float (&v)[CH]) const { v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; }
};
template <bool BF> MTT_DEV void store4(void* base, unsigned off, const float (&v)[CH]) {
  if (BF) *(u32x2*)((bf16_t*)base + off) = (u32x2){pack2(v[0], v[1]), pack2(v[2], v[3])};
  else *(float4*)((float*)base + off) = make_float4(v[0], v[1], v[2], v[3]);
}
template <bool BF> MTT_DEV void* at(void* p, int64_t elems) { return BF ? (void*)((bf16_t*)p + elems) : (void*)((float*)p + elems); }

// block -> (image, low-res row, column block): consecutive hardware block ids go round-robin over the 8 XCDs, so id % 8 picks the
// XCD and every image is given to exactly one of them.
MTT_DEV bool cell_of_block(const mtt_upconv_desc& d, unsigned gx, int& img, int& i, unsigned& xb) {
  const unsigned per_img = gx * (unsigned)d.h;
  const unsigned L = blockIdx.x, xcd = L & 7u, k = L >> 3;
  img = (int)((k / per_img) * 8u + xcd);
  const unsigned r = k % per_img;
  i = (int)(r / gx);
  xb = r - (unsigned)i * gx;
  return img < d.Z * d.B;
}

template <bool BF>
__global__ __launch_bounds__(256) void upconv4_expand_kernel(const mtt_upconv_desc d, unsigned gx) {
  int img, i; unsigned xb;
  if (!cell_of_block(d, gx, img, i, xb)) return;
  const unsigned CG = (unsigned)d.Cp / CH;
  const unsigned t = xb * 256u + threadIdx.x;
  if (t >= (unsigned)d.w * CG) return;
  const int j = (int)(t / CG), cg = (int)(t - (unsigned)j * CG);
  const int h = d.h, w = d.w;
  const unsigned Cp = (unsigned)d.Cp, zld = 9u * Cp;
  const unsigned cs[3] = {(unsigned)(j > 0 ? j - 1 : 0), (unsigned)j, (unsigned)(j < w - 1 ? j + 1 : w - 1)};
  const unsigned rs[3] = {(unsigned)(i > 0 ? i - 1 : 0), (unsigned)i, (unsigned)(i < h - 1 ? i + 1 : h - 1)};
  const float mx_lo = j > 0 ? 1.f : 0.f, mx_hi = j < w - 1 ? 1.f : 0.f;       // do hi-res columns 4j - 1 / 4j + 4 exist?
  const float my_lo = i > 0 ? 1.f : 0.f, my_hi = i < h - 1 ? 1.f : 0.f;
  const void* zimg = at<BF>(d.z, (int64_t)img * h * w * zld);

  float acc[4][4][CH];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[p][q][c] = 0.f;

  // A real loop over the tap row (27 neighbourhood loads in flight per trip; fully unrolled, the compiler hoists all 81 loads into one
  // burst and runs out of registers).  Its row-pass weights are then run-time, but block-uniform: scalar selects.
#pragma unroll 1
  for (int dy = 0; dy < 3; ++dy) {
    float R[4][3];                                        // weight of low-res row slot s (through tap row dy) on output row 4i + p
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int r1 = p + dy;                              // expansion row 4i + p + dy - 1  ->  table index r + 1
      const float m = r1 == 0 ? my_lo : (r1 == 5 ? my_hi : 1.f);
#pragma unroll
      for (int s = 0; s < 3; ++s) R[p][s] = (slot_a(r1) == s ? wgt_a(r1) : (slot_b(r1) == s ? wgt_b(r1) : 0.f)) * m;
    }
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      // tap row dy of low-res row slot s: column pass -> the 4 hi-res columns of this cell
      const unsigned rowoff = rs[s] * (unsigned)w * zld + (unsigned)(dy * 3) * Cp + (unsigned)cg * CH;
      Raw<BF> raw[3][3];
#pragma unroll
      for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int cslot = 0; cslot < 3; ++cslot) raw[dx][cslot].load(zimg, rowoff + cs[cslot] * zld + (unsigned)dx * Cp);
      float v[4][CH];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int c = 0; c < CH; ++c) v[q][c] = 0.f;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        float f[3][CH];
#pragma unroll
        for (int cslot = 0; cslot < 3; ++cslot) raw[dx][cslot].get(f[cslot]);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int r1 = q + dx;                                    // hi-res column 4j + q + dx - 1  ->  table index r + 1
          const float m = r1 == 0 ? mx_lo : (r1 == 5 ? mx_hi : 1.f);
          const float wa = wgt_a(r1) * m, wb = wgt_b(r1) * m;
#pragma unroll
          for (int c = 0; c < CH; ++c) v[q][c] = fmaf(wa, f[slot_a(r1)][c], fmaf(wb, f[slot_b(r1)][c], v[q][c]));
        }
      }
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int c = 0; c < CH; ++c) acc[p][q][c] = fmaf(R[p][s], v[q][c], acc[p][q][c]);
    }
  }

  const int zi = img / d.B;
  float bs[CH], sc[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int ch = cg * CH + c;
    bs[c] = (d.bias && ch < d.C) ? d.bias[(int64_t)zi * d.C + ch] : 0.f;
    sc[c] = (d.colscale && ch < d.C) ? d.colscale[(int64_t)zi * d.C + ch] : 1.f;
  }
  const unsigned W = 4u * (unsigned)w;
  void* yimg = at<BF>(d.y, (int64_t)img * 16 * h * w * Cp);
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const unsigned orow = ((unsigned)(4 * i + p) * W + 4u * (unsigned)j) * Cp + (unsigned)cg * CH;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float o[CH];
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        float x = fmaf(acc[p][q][c], sc[c], bs[c]);
        if (d.act == MTT_ACT_GELU) x = gelu_f(x);
        else if (d.act == MTT_ACT_RELU) x = fmaxf(x, 0.f);
        o[c] = x;
      }
      store4<BF>(yimg, orow + (unsigned)q * Cp, o);
    }
  }
}

template <bool BF>
__global__ __launch_bounds__(256) void upconv4_gather_kernel(const mtt_upconv_desc d, unsigned gx) {
  int img, i; unsigned xb;
  if (!cell_of_block(d, gx, img, i, xb)) return;
  const unsigned CG = (unsigned)d.Cp / CH;
  const unsigned t = xb * 256u + threadIdx.x;
  if (t >= (unsigned)d.w * CG) return;
  const int j = (int)(t / CG), cg = (int)(t - (unsigned)j * CG);
  const int h = d.h, w = d.w, H = 4 * h, W = 4 * w;
  const unsigned Cp = (unsigned)d.Cp;
  // weight of hi-res row 4i - 2 + m (m = 0..7) of the expansion on low-res row i; at a border row i also holds the replicated
  // neighbour's share, and rows outside the hi-res map do not exist
  float wy[8] = {0.125f, 0.375f, 0.625f, 0.875f, 0.875f, 0.625f, 0.375f, 0.125f};
  float wx[8] = {0.125f, 0.375f, 0.625f, 0.875f, 0.875f, 0.625f, 0.375f, 0.125f};
  if (i == 0) { wy[0] = 0.f; wy[1] = 0.f; wy[2] += 0.375f; wy[3] += 0.125f; }
  if (i == h - 1) { wy[6] = 0.f; wy[7] = 0.f; wy[4] += 0.125f; wy[5] += 0.375f; }
  if (j == 0) { wx[0] = 0.f; wx[1] = 0.f; wx[2] += 0.375f; wx[3] += 0.125f; }
  if (j == w - 1) { wx[6] = 0.f; wx[7] = 0.f; wx[4] += 0.125f; wx[5] += 0.375f; }
  const void* yimg = at<BF>(d.y, (int64_t)img * H * W * Cp);

  float acc[3][3][CH];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[a][b][c] = 0.f;

#pragma unroll
  for (int tr = 0; tr < 10; ++tr) {                       // output row rho = 4i - 3 + tr
    const int rho = 4 * i - 3 + tr;
    if (rho >= 0 && rho < H) {                            // workgroup-uniform
      const unsigned rowoff = (unsigned)rho * (unsigned)W * Cp + (unsigned)cg * CH;
      Raw<BF> raw[10];
#pragma unroll
      for (int u = 0; u < 10; ++u) {                      // output column X = 4j - 3 + u
        const int X = 4 * j - 3 + u;
        const bool ok = X >= 0 && X < W;
        raw[u].load(yimg, rowoff + (unsigned)(ok ? X : 4 * j) * Cp);
        if (!ok) raw[u].zero();
      }
      float s[3][CH];
#pragma unroll
      for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int c = 0; c < CH; ++c) s[dx][c] = 0.f;
#pragma unroll
      for (int u = 0; u < 10; ++u) {
        float f[CH];
        raw[u].get(f);
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const int n = u + dx - 2;                       // expansion column 4j - 2 + n = X + dx - 1
          if (n >= 0 && n < 8) {
#pragma unroll
            for (int c = 0; c < CH; ++c) s[dx][c] = fmaf(wx[n], f[c], s[dx][c]);
          }
        }
      }
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const int m = tr + dy - 2;                        // expansion row 4i - 2 + m = rho + dy - 1
        if (m >= 0 && m < 8) {
#pragma unroll
          for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[dy][dx][c] = fmaf(wy[m], s[dx][c], acc[dy][dx][c]);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  void* zimg = at<BF>(d.z, (int64_t)img * h * w * 9 * Cp);
  const unsigned zoff = ((unsigned)i * (unsigned)w + (unsigned)j) * 9u * Cp + (unsigned)cg * CH;
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) store4<BF>(zimg, zoff + (unsigned)(dy * 3 + dx) * Cp, acc[dy][dx]);
}

int check(const mtt_upconv_desc* d) {
  if (!d || !d->z || !d->y || d->Z <= 0 || d->B <= 0 || d->h <= 0 || d->w <= 0 || d->C <= 0 || d->C > d->Cp) return MTT_E_BADARG;
  if (d->Cp % 8) return MTT_E_ALIGN;
  if (d->z_dtype != d->y_dtype || (d->z_dtype != MTT_F32 && d->z_dtype != MTT_BF16)) return MTT_E_UNSUPPORTED;
  if ((int64_t)d->h * d->w * 16 * d->Cp * 4 >= (1LL << 31)) return MTT_E_UNSUPPORTED;      // 32-bit lane offsets inside one image
  return 0;
}

unsigned grid_of(const mtt_upconv_desc* d, unsigned& gx) {
  gx = ((unsigned)d->w * ((unsigned)d->Cp / CH) + 255u) / 256u;
  const unsigned imgs8 = ((unsigned)(d->Z * d->B) + 7u) / 8u;
  return imgs8 * 8u * gx * (unsigned)d->h;
}

}  // namespace

#define S_ ((hipStream_t)stream)

extern "C" int mtt_upconv4_expand(const mtt_upconv_desc* d, void* stream) {
  if (const int e = check(d)) return e;
  unsigned gx;
  const unsigned nb = grid_of(d, gx);
  if (d->z_dtype == MTT_BF16) hipLaunchKernelGGL(upconv4_expand_kernel<true>, dim3(nb), dim3(256), 0, S_, *d, gx);
  else hipLaunchKernelGGL(upconv4_expand_kernel<false>, dim3(nb), dim3(256), 0, S_, *d, gx);
  return (int)hipGetLastError();
}

extern "C" int mtt_upconv4_gather(const mtt_upconv_desc* d, void* stream) {
  if (const int e = check(d)) return e;
  unsigned gx;
  const unsigned nb = grid_of(d, gx);
  if (d->z_dtype == MTT_BF16) hipLaunchKernelGGL(upconv4_gather_kernel<true>, dim3(nb), dim3(256), 0, S_, *d, gx);
  else hipLaunchKernelGGL(upconv4_gather_kernel<false>, dim3(nb), dim3(256), 0, S_, *d, gx);
  return (int)hipGetLastError();
}
