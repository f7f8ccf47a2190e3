// HBM-bound row / elementwise kernels of the hot path: LayerNorm, row softmax, BatchNorm (training
// statistics), patchify, channel-attention logits, task-feature modulation, cross-task mixing,
// bilinear resize, casts and column sums.  All are coalesced over the channel (last) dimension with
// 8 channels (16 B of bf16 / 32 B of fp32) per lane; reductions over rows (BatchNorm, column sums) go through per-block
// partials in a caller-owned workspace that a second kernel merges in block order (deterministic, no atomics).
#include "mtt_device.h"

namespace {

MTT_DEV void ld8(const void* p, int64_t idx, int dtype, float (&v)[8]) {
  if (dtype == MTT_BF16) {
    const u32x4 u = *(const u32x4*)((const bf16_t*)p + idx);
#pragma unroll
    for (int q = 0; q < 4; ++q) { v[2 * q] = lo_of(u[q]); v[2 * q + 1] = hi_of(u[q]); }
  } else {
    const float4 a = *(const float4*)((const float*)p + idx);
    const float4 b = *(const float4*)((const float*)p + idx + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
}
MTT_DEV void st8(void* p, int64_t idx, int dtype, const float (&v)[8]) {
  if (dtype == MTT_BF16) {
    *(u32x4*)((bf16_t*)p + idx) = (u32x4){pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
  } else {
    *(float4*)((float*)p + idx) = make_float4(v[0], v[1], v[2], v[3]);
    *(float4*)((float*)p + idx + 4) = make_float4(v[4], v[5], v[6], v[7]);
  }
}

// 8 values as MTT_F32 / MTT_BF16, or as MTT_SPLIT planes: hi = bf16(v) at p, lo = bf16(v - hi) at p_lo
MTT_DEV void st8s(void* p, void* p_lo, int64_t idx, int dtype, const float (&v)[8]) {
  if (dtype == MTT_SPLIT) {
    const u32x4 hi = (u32x4){pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
    *(u32x4*)((bf16_t*)p + idx) = hi;
    *(u32x4*)((bf16_t*)p_lo + idx) = (u32x4){pack2(v[0] - lo_of(hi.x), v[1] - hi_of(hi.x)), pack2(v[2] - lo_of(hi.y), v[3] - hi_of(hi.y)),
                                             pack2(v[4] - lo_of(hi.z), v[5] - hi_of(hi.z)), pack2(v[6] - lo_of(hi.w), v[7] - hi_of(hi.w))};
  } else {
    st8(p, idx, dtype, v);
  }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, 4 rows per block.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ln_fwd_kernel(const mtt_ln_desc d) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= d.rows) return;
  const float* x = d.x + row * d.ldx;
  const int C4 = d.C >> 2;
  float s = 0.f;
  for (int c = lane; c < C4; c += 64) { const float4 v = ((const float4*)x)[c]; s += (v.x + v.y) + (v.z + v.w); }
  const float mean = wave_sum(s) / d.C;
  float q = 0.f;
  for (int c = lane; c < C4; c += 64) {
    const float4 v = ((const float4*)x)[c];
    const float a = v.x - mean, b = v.y - mean, e = v.z - mean, f = v.w - mean;
    q += (a * a + b * b) + (e * e + f * f);
  }
  const float rstd = rsqrtf(wave_sum(q) / d.C + d.eps);
  if (lane == 0) { if (d.mean) d.mean[row] = mean; if (d.rstd) d.rstd[row] = rstd; }
  for (int c = lane; c < C4; c += 64) {
    const float4 v = ((const float4*)x)[c];
    const float4 g = ((const float4*)d.gamma)[c];
    const float4 bb = ((const float4*)d.beta)[c];
    const float y0 = (v.x - mean) * rstd * g.x + bb.x, y1 = (v.y - mean) * rstd * g.y + bb.y;
    const float y2 = (v.z - mean) * rstd * g.z + bb.z, y3 = (v.w - mean) * rstd * g.w + bb.w;
    const int64_t o = row * d.ldy + (int64_t)c * 4;
    if (d.y_dtype == MTT_F32) *(float4*)((float*)d.y + o) = make_float4(y0, y1, y2, y3);
    else {
      const u32x2 hi = (u32x2){pack2(y0, y1), pack2(y2, y3)};
      *(u32x2*)((bf16_t*)d.y + o) = hi;
      if (d.y_dtype == MTT_SPLIT)          // x = hi + lo: the lo plane carries the bf16 rounding residual (fp32-class operand for the LDS-DMA x3 GEMM)
        *(u32x2*)((bf16_t*)d.y_lo + o) = (u32x2){pack2(y0 - lo_of(hi.x), y1 - hi_of(hi.x)), pack2(y2 - lo_of(hi.y), y3 - hi_of(hi.y))};
    }
    if (d.y32) *(float4*)(d.y32 + row * d.ldy32 + (int64_t)c * 4) = make_float4(y0, y1, y2, y3);
  }
}

// C <= 1024: the row lives in registers (4 float4 chunks per lane) between the statistics and the store — ONE read of x instead of three —
// gamma / beta are loaded once per wave and a wave walks rows with a grid stride (the kernel above spends a wave per row: 16 k workgroups
// of 4 rows for one encoder LayerNorm).  Same summation order as above: bitwise the same statistics and outputs.
__global__ __launch_bounds__(256) void ln_fwd_reg_kernel(const mtt_ln_desc d) {
  const int lane = threadIdx.x & 63;
  const int C4 = d.C >> 2;
  float4 ga[4], be[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c4 = lane + 64 * k;
    ga[k] = c4 < C4 ? ((const float4*)d.gamma)[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
    be[k] = c4 < C4 ? ((const float4*)d.beta)[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int64_t stride = (int64_t)gridDim.x * 4;
  for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < d.rows; row += stride) {
    const float4* x = (const float4*)(d.x + row * d.ldx);
    float4 v[4];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c4 = lane + 64 * k;
      v[k] = c4 < C4 ? x[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
      if (c4 < C4) s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    }
    const float mean = wave_sum(s) / d.C;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (lane + 64 * k < C4) {
        const float a = v[k].x - mean, b = v[k].y - mean, e = v[k].z - mean, f = v[k].w - mean;
        q += (a * a + b * b) + (e * e + f * f);
      }
    }
    const float rstd = rsqrtf(wave_sum(q) / d.C + d.eps);
    if (lane == 0) { if (d.mean) d.mean[row] = mean; if (d.rstd) d.rstd[row] = rstd; }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c4 = lane + 64 * k;
      if (c4 >= C4) continue;
      const float y0 = (v[k].x - mean) * rstd * ga[k].x + be[k].x, y1 = (v[k].y - mean) * rstd * ga[k].y + be[k].y;
      const float y2 = (v[k].z - mean) * rstd * ga[k].z + be[k].z, y3 = (v[k].w - mean) * rstd * ga[k].w + be[k].w;
      const int64_t o = row * d.ldy + (int64_t)c4 * 4;
      if (d.y_dtype == MTT_F32) *(float4*)((float*)d.y + o) = make_float4(y0, y1, y2, y3);
      else {
        const u32x2 hi = (u32x2){pack2(y0, y1), pack2(y2, y3)};
        *(u32x2*)((bf16_t*)d.y + o) = hi;
        if (d.y_dtype == MTT_SPLIT)
          *(u32x2*)((bf16_t*)d.y_lo + o) = (u32x2){pack2(y0 - lo_of(hi.x), y1 - hi_of(hi.x)), pack2(y2 - lo_of(hi.y), y3 - hi_of(hi.y))};
      }
      if (d.y32) *(float4*)(d.y32 + row * d.ldy32 + (int64_t)c4 * 4) = make_float4(y0, y1, y2, y3);
    }
  }
}

// backward, part 1 (one wave per row): dx += rstd * (g - mean(g) - xhat * mean(g*xhat)), g = dy*gamma
__global__ __launch_bounds__(256) void ln_bwd_dx_kernel(const mtt_ln_desc d) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= d.rows) return;
  const float* x = d.x + row * d.ldx;
  const float mean = d.mean[row], rstd = d.rstd[row];
  float s1 = 0.f, s2 = 0.f;
  for (int c = lane; c < d.C; c += 64) {
    const float xh = (x[c] - mean) * rstd;
    const float g = ld_elem(d.dy, row * d.ldy + c, d.y_dtype) * d.gamma[c];
    s1 += g; s2 += g * xh;
  }
  s1 = wave_sum(s1) / d.C; s2 = wave_sum(s2) / d.C;
  float* dx = d.dx + row * d.ldx;
  const float* din = (d.dx_in ? d.dx_in : d.dx) + row * d.ldx;
  for (int c = lane; c < d.C; c += 64) {
    const float xh = (x[c] - mean) * rstd;
    const float g = ld_elem(d.dy, row * d.ldy + c, d.y_dtype) * d.gamma[c];
    dx[c] = din[c] + rstd * (g - s1 - xh * s2);
  }
}

// backward, fused (C <= 1024): one read of x and dy per row.  A wave owns a row at a time (4 float4 chunks per lane kept in
// registers between the statistics pass and the dx pass) and keeps per-column partial sums of dgamma / dbeta for all its rows;
// the 4 waves are combined through LDS and the block writes one partial row to the workspace (merged by ln_dgb_final_kernel).
__global__ __launch_bounds__(256) void ln_bwd_fused_kernel(const mtt_ln_desc d, int rows_per_block) {
  __shared__ float part[4][2][1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int C4 = d.C >> 2;
  float4 ga[4], ag[4], ab[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c4 = lane + 64 * k;
    ga[k] = c4 < C4 ? ((const float4*)d.gamma)[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
    ag[k] = make_float4(0.f, 0.f, 0.f, 0.f); ab[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < d.rows ? r0 + rows_per_block : d.rows;
  const float invC = 1.0f / (float)d.C;
  for (int64_t row = r0 + wave; row < r1; row += 4) {
    const float mean = d.mean[row], rstd = d.rstd[row];
    float4 xh[4], g[4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c4 = lane + 64 * k;
      xh[k] = make_float4(0.f, 0.f, 0.f, 0.f); g[k] = xh[k];
      if (c4 < C4) {
        const float4 xv = *(const float4*)(d.x + row * d.ldx + c4 * 4);
        float4 dy;
        if (d.y_dtype == MTT_F32) dy = *(const float4*)((const float*)d.dy + row * d.ldy + c4 * 4);
        else {
          const u32x2 r = *(const u32x2*)((const bf16_t*)d.dy + row * d.ldy + c4 * 4);
          dy = make_float4(lo_of(r[0]), hi_of(r[0]), lo_of(r[1]), hi_of(r[1]));
        }
        xh[k] = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd);
        ag[k].x += dy.x * xh[k].x; ag[k].y += dy.y * xh[k].y; ag[k].z += dy.z * xh[k].z; ag[k].w += dy.w * xh[k].w;
        ab[k].x += dy.x; ab[k].y += dy.y; ab[k].z += dy.z; ab[k].w += dy.w;
        g[k] = make_float4(dy.x * ga[k].x, dy.y * ga[k].y, dy.z * ga[k].z, dy.w * ga[k].w);
        s1 += g[k].x + g[k].y + g[k].z + g[k].w;
        s2 += g[k].x * xh[k].x + g[k].y * xh[k].y + g[k].z * xh[k].z + g[k].w * xh[k].w;
      }
    }
    s1 = wave_sum(s1) * invC; s2 = wave_sum(s2) * invC;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c4 = lane + 64 * k;
      if (c4 < C4) {
        float4* dx = (float4*)(d.dx + row * d.ldx + c4 * 4);
        float4 o = *(const float4*)((d.dx_in ? d.dx_in : d.dx) + row * d.ldx + c4 * 4);
        o.x += rstd * (g[k].x - s1 - xh[k].x * s2); o.y += rstd * (g[k].y - s1 - xh[k].y * s2);
        o.z += rstd * (g[k].z - s1 - xh[k].z * s2); o.w += rstd * (g[k].w - s1 - xh[k].w * s2);
        *dx = o;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c4 = lane + 64 * k;
    if (c4 < C4) { *(float4*)&part[wave][0][c4 * 4] = ag[k]; *(float4*)&part[wave][1][c4 * 4] = ab[k]; }
  }
  __syncthreads();
  float* wsb = d.ws + (int64_t)blockIdx.x * 2 * d.C;
  for (int c = threadIdx.x; c < d.C; c += 256) {
    wsb[c] = (part[0][0][c] + part[1][0][c]) + (part[2][0][c] + part[3][0][c]);
    wsb[d.C + c] = (part[0][1][c] + part[1][1][c]) + (part[2][1][c] + part[3][1][c]);
  }
}

// dgamma / dbeta = the per-block partials of ws [nblk][2][C] summed in block order (deterministic; 32 columns x 8 partial lanes).
// grid (C / 32, 2): blockIdx.y = 0 sums the dgamma partials, 1 the dbeta partials; 8 row groups x 4 independent accumulators per lane keep
// 32 loads in flight per lane (the one-accumulator form ran the 8 MB of partials of an encoder LayerNorm at 0.2 TB/s from 32 workgroups);
// fixed summation order.
__global__ __launch_bounds__(256) void ln_dgb_final_kernel(const float* ws, float* dgamma, float* dbeta, int C, int nblk) {
  __shared__ float sh[8][32];
  const int cl = threadIdx.x & 31, pl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  const float* src = ws + (blockIdx.y ? C : 0) + c;
  float t[4] = {0.f, 0.f, 0.f, 0.f};
  if (c < C) {
    int b = pl;
    for (; b + 24 < nblk; b += 32) {
#pragma unroll
      for (int u = 0; u < 4; ++u) t[u] += src[(int64_t)(b + 8 * u) * 2 * C];
    }
    for (; b < nblk; b += 8) t[0] += src[(int64_t)b * 2 * C];
  }
  float v = (t[0] + t[1]) + (t[2] + t[3]);
  sh[pl][cl] = v;
  __syncthreads();
  if (pl == 0 && c < C) {
#pragma unroll
    for (int l = 1; l < 8; ++l) v += sh[l][cl];
    (blockIdx.y ? dbeta : dgamma)[c] = v;
  }
}

// backward, part 2 (column-parallel, any C): per-block partial sums of dy*xhat and dy.  Thread = one column group of 4 (and, for
// C < 1024, one of 256 / (C/4) row lanes, combined through LDS in lane order); one partial row per block goes to ws [nblk][2][C].
__global__ __launch_bounds__(256) void ln_bwd_dgb_kernel(const mtt_ln_desc d, int rows_per_block) {
  extern __shared__ float lsm[];           // [lanes][2][C]
  const int C4 = d.C >> 2;
  const int lanes = C4 < 256 ? 256 / C4 : 1;
  const int rl = C4 < 256 ? threadIdx.x / C4 : 0;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < d.rows ? r0 + rows_per_block : d.rows;
  if (rl < lanes) {
    for (int c4 = threadIdx.x % (C4 < 256 ? C4 : 256); c4 < C4; c4 += 256) {
      float ag[4] = {0.f, 0.f, 0.f, 0.f}, ab[4] = {0.f, 0.f, 0.f, 0.f};
      for (int64_t r = r0 + rl; r < r1; r += lanes) {
        const float mean = d.mean[r], rstd = d.rstd[r];
        const float4 xv = *(const float4*)(d.x + r * d.ldx + c4 * 4);
        float dy[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) dy[j] = ld_elem(d.dy, r * d.ldy + c4 * 4 + j, d.y_dtype);
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) { ag[j] += dy[j] * (xs[j] - mean) * rstd; ab[j] += dy[j]; }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) { lsm[(rl * 2) * d.C + c4 * 4 + j] = ag[j]; lsm[(rl * 2 + 1) * d.C + c4 * 4 + j] = ab[j]; }
    }
  }
  __syncthreads();
  float* wsb = d.ws + (int64_t)blockIdx.x * 2 * d.C;
  for (int c = threadIdx.x; c < d.C; c += 256) {
    float t0 = lsm[c], t1 = lsm[d.C + c];
    for (int l = 1; l < lanes; ++l) { t0 += lsm[(l * 2) * d.C + c]; t1 += lsm[(l * 2 + 1) * d.C + c]; }
    wsb[c] = t0; wsb[d.C + c] = t1;
  }
}

// ------------------------------------------------------------------------------------------------
// Row softmax on a materialised matrix (InvPT decoder attention, round-1 attention backward).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const mtt_softmax_desc d) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= d.rows) return;
  const int64_t base = row * d.ld;
  const int nch = (int)(d.ld >> 3);
  float mx = -INFINITY;
  for (int ch = lane; ch < nch; ch += 64) {
    float v[8];
    ld8(d.S, base + ch * 8, d.s_dtype, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) if (ch * 8 + j < d.cols) mx = fmaxf(mx, v[j] * d.scale);
  }
  mx = wave_max(mx);
  float s = 0.f;
  for (int ch = lane; ch < nch; ch += 64) {
    float v[8];
    ld8(d.S, base + ch * 8, d.s_dtype, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) if (ch * 8 + j < d.cols) s += __expf(v[j] * d.scale - mx);
  }
  const float inv = 1.0f / wave_sum(s);
  for (int ch = lane; ch < nch; ch += 64) {
    float v[8];
    ld8(d.S, base + ch * 8, d.s_dtype, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = ch * 8 + j < d.cols ? __expf(v[j] * d.scale - mx) * inv : 0.f;   // padding columns -> 0
    st8(d.P, base + ch * 8, d.p_dtype, v);
  }
}

// dS = scale * P * (dP - sum(dP*P))  (+ extra[r, c] for the first extra_rows rows of every matrix)
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const mtt_softmax_desc d) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= d.rows) return;
  const int64_t base = row * d.ld;
  const int nch = (int)(d.ld >> 3);
  float s = 0.f;
  for (int ch = lane; ch < nch; ch += 64) {
    float pv[8], gv[8];
    ld8(d.P, base + ch * 8, d.p_dtype, pv);
    ld8(d.dP, base + ch * 8, d.s_dtype, gv);
#pragma unroll
    for (int j = 0; j < 8; ++j) if (ch * 8 + j < d.cols) s += pv[j] * gv[j];
  }
  s = wave_sum(s);
  const int64_t mat = d.rows_per_mat > 0 ? row / d.rows_per_mat : 0;
  const int64_t rin = d.rows_per_mat > 0 ? row - mat * d.rows_per_mat : row;
  const bool ex = d.extra != nullptr && rin < d.extra_rows;
  for (int ch = lane; ch < nch; ch += 64) {
    float pv[8], gv[8];
    ld8(d.P, base + ch * 8, d.p_dtype, pv);
    ld8(d.dP, base + ch * 8, d.s_dtype, gv);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = ch * 8 + j;
      float v = 0.f;
      if (c < d.cols) {
        v = d.scale * pv[j] * (gv[j] - s);
        if (ex) v += d.extra[(mat * d.extra_rows + rin) * d.extra_ld + c];
      }
      gv[j] = v;
    }
    st8(d.dS, base + ch * 8, d.s_dtype, gv);
  }
}

// ------------------------------------------------------------------------------------------------
// Patchify: img fp32 NCHW -> cols [B*h*w, 768], k = (c*16 + dy)*16 + dx.  One thread = 8 dx.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void patchify_kernel(const float* img, void* cols, int B, int H, int W, int out_dtype) {
  const int h = H >> 4, w = W >> 4;
  const int64_t total = (int64_t)B * h * w * 96;   // 768 / 8 chunks per patch
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int chunk = (int)(t % 96);
    const int64_t patch = t / 96;
    const int px = (int)(patch % w), py = (int)((patch / w) % h), b = (int)(patch / ((int64_t)w * h));
    const int c = chunk >> 5, dy = (chunk >> 1) & 15, dx0 = (chunk & 1) * 8;
    float v[8];
    ld8(img, (((int64_t)b * 3 + c) * H + py * 16 + dy) * W + px * 16 + dx0, MTT_F32, v);
    st8(cols, patch * 768 + chunk * 8, out_dtype, v);
  }
}

// ------------------------------------------------------------------------------------------------
// Channel-attention logits: rawchan[b,t,win,c] += sum_{p in win, p in split} q[b,t,p] * xn[b,T+p,c]
// grid (C/8/32 column groups, pixel splits, B*nwin); block = 32 column chunks x 8 pixel lanes.
// ------------------------------------------------------------------------------------------------
constexpr int CL_MAXT = 8;
// the normalised tokens of the forward: one storage type, or MTT_SPLIT = hi + lo bf16 planes (q is then fp32)
MTT_DEV void cl_ld8_x(const mtt_chanlogit_desc& d, int64_t off, float (&v)[8]) {
  if (d.dtype == MTT_SPLIT) {
    float lo[8];
    ld8(d.xn, off, MTT_BF16, v);
    ld8(d.xn_lo, off, MTT_BF16, lo);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] += lo[j];
  } else {
    ld8(d.xn, off, d.dtype, v);
  }
}
MTT_DEV int cl_qdtype(const mtt_chanlogit_desc& d) { return d.dtype == MTT_SPLIT ? MTT_F32 : d.dtype; }
// grid (C/64 column groups, pixel splits, B*nwin); block = 8 column chunks (64 channels, 128 B per pixel) x 32 pixel lanes.
// Pixel lanes are reduced with wave shuffles, the 4 waves through LDS; gridDim.y == 1 -> the result is stored, else pixel split s
// stores its partial sums as plane s of the workspace (summed in split order by mtt_reduce_few_kernel: deterministic, no atomics).
__global__ __launch_bounds__(256) void chanlogit_kernel(const mtt_chanlogit_desc d, int tbase, float* part, int64_t plane_elems) {
  const int nwin = d.nh * d.nw, wh = d.h / d.nh, ww = d.w / d.nw, P = wh * ww;
  const int b = blockIdx.z / nwin, win = blockIdx.z % nwin;
  const int wy = win / d.nw, wx = win % d.nw;
  const int cl = threadIdx.x & 7;
  const int cchunk = blockIdx.x * 8 + cl;
  const int plane = threadIdx.x >> 3;                 // 0..31
  const int nT = d.T - tbase < CL_MAXT ? d.T - tbase : CL_MAXT;
  float acc[CL_MAXT][8];
#pragma unroll
  for (int t = 0; t < CL_MAXT; ++t)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[t][j] = 0.f;
  const bool cok = cchunk * 8 < d.C;
  const int per = (P + gridDim.y - 1) / gridDim.y;
  const int p0 = blockIdx.y * per, p1 = p0 + per < P ? p0 + per : P;
  if (cok) {
#pragma unroll 2
    for (int pi = p0 + plane; pi < p1; pi += 32) {
      const int y = wy * wh + pi / ww, x = wx * ww + pi % ww;
      const int pix = y * d.w + x;
      float xv[8];
      cl_ld8_x(d, ((int64_t)b * d.N + d.T + pix) * d.C + cchunk * 8, xv);
#pragma unroll
      for (int t = 0; t < CL_MAXT; ++t) {
        if (t < nT) {
          const float qv = ld_elem(d.q, ((int64_t)b * d.T + tbase + t) * d.ldq + pix, cl_qdtype(d));
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[t][j] += qv * xv[j];
        }
      }
    }
  }
  __shared__ float red[4][CL_MAXT][64];
  const int wave = threadIdx.x >> 6;
#pragma unroll
  for (int t = 0; t < CL_MAXT; ++t) {
    if (t < nT) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float v = acc[t][j];
        v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
        if ((threadIdx.x & 63) < 8) red[wave][t][cl * 8 + j] = v;
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nT * 64; i += 256) {
    const int t = i >> 6, c = i & 63;
    const int col = blockIdx.x * 64 + c;
    if (col >= d.C) continue;
    const float v = red[0][t][c] + red[1][t][c] + red[2][t][c] + red[3][t][c];
    const int64_t oi = (((int64_t)b * d.T + tbase + t) * nwin + win) * d.C + col;
    if (gridDim.y == 1) d.rawchan[oi] = v; else part[(int64_t)blockIdx.y * plane_elems + oi] = v;
  }
}

// Same contract, EIGHT consecutive pixels per lane and iteration (windows whose width is a multiple of 8, ldq % 8 == 0): the T query
// values of those pixels come as one 16-byte load per task instead of one scalar load per (task, pixel) — the one-pixel form above
// issues 7 loads per 48 FMAs and ran at 1.3 TB/s on the [B, N, C] tokens; this one issues 14 per 384.
__global__ __launch_bounds__(256) void chanlogit_px8_kernel(const mtt_chanlogit_desc d, int tbase, float* part, int64_t plane_elems) {
  const int nwin = d.nh * d.nw, wh = d.h / d.nh, ww = d.w / d.nw, P = wh * ww;
  const int b = blockIdx.z / nwin, win = blockIdx.z % nwin;
  const int wy = win / d.nw, wx = win % d.nw;
  const int cl = threadIdx.x & 7;
  const int cchunk = blockIdx.x * 8 + cl;
  const int plane = threadIdx.x >> 3;                 // 0..31
  const int nT = d.T - tbase < CL_MAXT ? d.T - tbase : CL_MAXT;
  float acc[CL_MAXT][8];
#pragma unroll
  for (int t = 0; t < CL_MAXT; ++t)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[t][j] = 0.f;
  const bool cok = cchunk * 8 < d.C;
  const int per = ((P + gridDim.y - 1) / gridDim.y + 7) & ~7;
  const int p0 = blockIdx.y * per, p1 = p0 + per < P ? p0 + per : P;
  if (cok) {
    for (int pi = p0 + plane * 8; pi < p1; pi += 256) {
      const int y = wy * wh + pi / ww, x = wx * ww + pi % ww;
      const int pix = y * d.w + x;                      // pixels pix .. pix + 7 lie in one row of the window
      float qv[CL_MAXT][8];
#pragma unroll
      for (int t = 0; t < CL_MAXT; ++t)
        if (t < nT) ld8(d.q, ((int64_t)b * d.T + tbase + t) * d.ldq + pix, cl_qdtype(d), qv[t]);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float xv[8];
        cl_ld8_x(d, ((int64_t)b * d.N + d.T + pix + k) * d.C + cchunk * 8, xv);
#pragma unroll
        for (int t = 0; t < CL_MAXT; ++t)
          if (t < nT) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[t][j] += qv[t][k] * xv[j];
          }
      }
    }
  }
  __shared__ float red[4][CL_MAXT][64];
  const int wave = threadIdx.x >> 6;
#pragma unroll
  for (int t = 0; t < CL_MAXT; ++t) {
    if (t < nT) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float v = acc[t][j];
        v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
        if ((threadIdx.x & 63) < 8) red[wave][t][cl * 8 + j] = v;
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nT * 64; i += 256) {
    const int t = i >> 6, c = i & 63;
    const int col = blockIdx.x * 64 + c;
    if (col >= d.C) continue;
    const float v = red[0][t][c] + red[1][t][c] + red[2][t][c] + red[3][t][c];
    const int64_t oi = (((int64_t)b * d.T + tbase + t) * nwin + win) * d.C + col;
    if (gridDim.y == 1) d.rawchan[oi] = v; else part[(int64_t)blockIdx.y * plane_elems + oi] = v;
  }
}

// ------------------------------------------------------------------------------------------------
// Task-feature modulation (taskprompter.py:436-467).  One thread = 8 channels of one token.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void modulate_kernel(const mtt_modulate_desc d) {
  const int hw = d.h * d.w, C8 = d.C >> 3, nwin = d.nh * d.nw, wh = d.h / d.nh, ww = d.w / d.nw;
  const int hg = d.hg > 0 ? d.hg : 64;
  const int nH = d.C / hg;
  const int64_t total = (int64_t)d.B * hw * C8;
  const int64_t plane = (int64_t)d.B * hw * d.C;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(t % C8);
    const int64_t tok = t / C8;
    const int p = (int)(tok % hw), b = (int)(tok / hw);
    const int y = p / d.w, x = p % d.w;
    const int win = (y / wh) * d.nw + (x / ww);
    float xv[8], ov[8];
    ld8(d.x, (int64_t)b * d.x_bs + (int64_t)p * d.x_ld + c8 * 8, MTT_F32, xv);
    const int head = (c8 * 8) / hg;
    for (int tk = 0; tk < d.T; ++tk) {
      const float a = d.rawlog[(((int64_t)b * nH + head) * d.T + tk) * d.N + d.T + p];
#pragma unroll
      for (int j = 0; j < 8; ++j) ov[j] = xv[j] * (1.0f + a);
      st8s(d.out, d.out_lo, (int64_t)(2 * tk) * plane + tok * d.C + c8 * 8, d.out_dtype, ov);
      float bw[8];
      ld8(d.rawchan, (((int64_t)b * d.T + tk) * nwin + win) * d.C + c8 * 8, MTT_F32, bw);
#pragma unroll
      for (int j = 0; j < 8; ++j) ov[j] = xv[j] * (1.0f + bw[j]);
      st8s(d.out, d.out_lo, (int64_t)(2 * tk + 1) * plane + tok * d.C + c8 * 8, d.out_dtype, ov);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Cross-task mixing: out[t][row,:] (+)= sum_s wmix[b,t,s] * fea[s][row,:]
// ------------------------------------------------------------------------------------------------
constexpr int CTR_MAXT = 8;
__global__ __launch_bounds__(256) void ctr_mix_kernel(const mtt_ctr_desc d) {
  const int C8 = (d.C + 7) >> 3;
  const int64_t rows = (int64_t)d.B * d.rows_per_b;
  const int64_t total = rows * C8;
  const int64_t plane = rows * d.ld;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(i % C8);
    const int64_t row = i / C8;
    const int b = (int)(row / d.rows_per_b);
    float f[CTR_MAXT][8];
#pragma unroll
    for (int s = 0; s < CTR_MAXT; ++s)
      if (s < d.T) ld8(d.fea, (int64_t)s * plane + row * d.ld + c8 * 8, d.fea_dtype, f[s]);
    for (int t = 0; t < d.T; ++t) {
      float o[8];
      const int64_t oi = (int64_t)t * plane + row * d.ld + c8 * 8;
      if (d.accumulate) ld8(d.out, oi, MTT_F32, o);
      else {
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = 0.f;
      }
#pragma unroll
      for (int s = 0; s < CTR_MAXT; ++s)
        if (s < d.T) {
          const float w = d.wmix[((int64_t)b * d.T + t) * d.T + s];
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] += w * f[s][j];
        }
      st8(d.out, oi, d.accumulate ? MTT_F32 : d.out_dtype, o);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Bilinear resize, align_corners = False (PyTorch semantics: src = (o + .5) * in/out - .5, clamped at 0).
// ------------------------------------------------------------------------------------------------
MTT_DEV void src_index(int o, int in, int out, int& i0, int& i1, float& w1) {
  const float scale = (float)in / (float)out;
  float s = ((float)o + 0.5f) * scale - 0.5f;
  if (s < 0.f) s = 0.f;
  i0 = (int)s;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 < in - 1 ? i0 + 1 : i0;
  w1 = s - (float)i0;
}

// grid (column blocks, output row, batch): the row / batch come from the block indices and the column index is split with one
// 32-bit division — the flat 64-bit div/mod chain per 16-byte output chunk cost more than the interpolation itself.
__global__ __launch_bounds__(256) void bilinear_fwd_nhwc_kernel(const mtt_resize_desc d) {
  const unsigned C8 = (unsigned)((d.C + 7) >> 3);
  const int oy = blockIdx.y, b = blockIdx.z;
  int y0, y1; float wy;
  src_index(oy, d.Hin, d.Hout, y0, y1, wy);
  const int64_t ib = (int64_t)b * d.Hin * d.Win;
  const int64_t r0 = (ib + (int64_t)y0 * d.Win) * d.ld_in, r1 = (ib + (int64_t)y1 * d.Win) * d.ld_in;
  const int64_t orow = ((int64_t)b * d.Hout + oy) * d.Wout;
  const unsigned total = (unsigned)d.Wout * C8;
  for (unsigned t = blockIdx.x * 256u + threadIdx.x; t < total; t += gridDim.x * 256u) {
    const unsigned ox = t / C8, c8 = t - ox * C8;
    int x0, x1; float wx;
    src_index((int)ox, d.Win, d.Wout, x0, x1, wx);
    float v00[8], v01[8], v10[8], v11[8], o[8];
    ld8(d.in, r0 + (int64_t)x0 * d.ld_in + c8 * 8, d.in_dtype, v00);
    ld8(d.in, r0 + (int64_t)x1 * d.ld_in + c8 * 8, d.in_dtype, v01);
    ld8(d.in, r1 + (int64_t)x0 * d.ld_in + c8 * 8, d.in_dtype, v10);
    ld8(d.in, r1 + (int64_t)x1 * d.ld_in + c8 * 8, d.in_dtype, v11);
    const int64_t oi = (orow + ox) * d.ld_out + c8 * 8;
    if (d.accumulate) ld8(d.out, oi, d.out_dtype, o);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float top = v00[j] * (1.f - wx) + v01[j] * wx, bot = v10[j] * (1.f - wx) + v11[j] * wx;
      const float r = top * (1.f - wy) + bot * wy;
      o[j] = d.accumulate ? o[j] + r : r;
    }
    st8(d.out, oi, d.out_dtype, o);
  }
}

// NHWC in -> NCHW fp32 out (the module's output contract, taskprompter_wrapper.py:36); grid (column blocks, output row, batch)
__global__ __launch_bounds__(256) void bilinear_fwd_nchw_kernel(const mtt_resize_desc d) {
  const int oy = blockIdx.y, b = blockIdx.z;
  int y0, y1; float wy;
  src_index(oy, d.Hin, d.Hout, y0, y1, wy);
  const int64_t ib = (int64_t)b * d.Hin * d.Win;
  for (int ox = blockIdx.x * 256 + threadIdx.x; ox < d.Wout; ox += gridDim.x * 256) {
    int x0, x1; float wx;
    src_index(ox, d.Win, d.Wout, x0, x1, wx);
    const int64_t a00 = (ib + (int64_t)y0 * d.Win + x0) * d.ld_in, a01 = (ib + (int64_t)y0 * d.Win + x1) * d.ld_in;
    const int64_t a10 = (ib + (int64_t)y1 * d.Win + x0) * d.ld_in, a11 = (ib + (int64_t)y1 * d.Win + x1) * d.ld_in;
    float* out = (float*)d.out + ((int64_t)b * d.C * d.Hout + oy) * d.Wout + ox;
    for (int c = 0; c < d.C; ++c) {
      const float top = ld_elem(d.in, a00 + c, d.in_dtype) * (1.f - wx) + ld_elem(d.in, a01 + c, d.in_dtype) * wx;
      const float bot = ld_elem(d.in, a10 + c, d.in_dtype) * (1.f - wx) + ld_elem(d.in, a11 + c, d.in_dtype) * wx;
      out[(int64_t)c * d.Hout * d.Wout] = top * (1.f - wy) + bot * wy;
    }
  }
}

// Integer scale S (2 or 4: the x4 / x2 resize of the head predictions to the image size, taskprompter_wrapper.py:36): one lane = one INPUT
// column j of output row oy and 4 channels -> its S output columns S j .. S j + S - 1 of each channel.  The three source columns j - 1, j,
// j + 1 (clamped) are loaded once for all S outputs (6 loads of 4 channels per 4 S outputs instead of 4 scalar loads per output) and each
// channel's outputs leave as one 16 / 8-byte store.  Weights are the exact constants of src_index for an integer scale.
MTT_DEV void ld4(const void* p, int64_t idx, int dtype, float (&v)[4]) {      // idx % 4 == 0, 16 / 8-byte aligned
  if (dtype == MTT_F32) { const float4 q = *(const float4*)((const float*)p + idx); v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; }
  else { const uint2 q = *(const uint2*)((const bf16_t*)p + idx); v[0] = lo_of(q.x); v[1] = hi_of(q.x); v[2] = lo_of(q.y); v[3] = hi_of(q.y); }
}
template <int S>
__global__ __launch_bounds__(256) void bilinear_fwd_nchw_int_kernel(const mtt_resize_desc d) {
  const int oy = blockIdx.y, b = blockIdx.z;
  int y0, y1; float wy;
  src_index(oy, d.Hin, d.Hout, y0, y1, wy);
  const int64_t ib = (int64_t)b * d.Hin * d.Win;
  const unsigned C4 = (unsigned)((d.C + 3) >> 2), total = (unsigned)d.Win * C4;
  for (unsigned t = blockIdx.x * 256u + threadIdx.x; t < total; t += gridDim.x * 256u) {
    const int c4 = (int)(t / (unsigned)d.Win), j = (int)(t - (unsigned)c4 * (unsigned)d.Win);   // lanes run along the row: coalesced stores
    const int xm = j > 0 ? j - 1 : 0, xp = j < d.Win - 1 ? j + 1 : j;
    float r0[3][4], r1[3][4];
    const int xs[3] = {xm, j, xp};
#pragma unroll
    for (int k = 0; k < 3; ++k) {                    // 4 channels per load (the pitch is pad8(C) >= 4 C4: padding channels are readable zeros)
      ld4(d.in, (ib + (int64_t)y0 * d.Win + xs[k]) * d.ld_in + c4 * 4, d.in_dtype, r0[k]);
      ld4(d.in, (ib + (int64_t)y1 * d.Win + xs[k]) * d.ld_in + c4 * 4, d.in_dtype, r1[k]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = c4 * 4 + q;
      if (c >= d.C) break;
      float o[S];
#pragma unroll
      for (int r = 0; r < S; ++r) {
        const int k0 = r < S / 2 ? 0 : 1;                                     // source pair (j - 1, j) or (j, j + 1)
        const float wx = r < S / 2 ? (r + 0.5f) / S + 0.5f : (r + 0.5f) / S - 0.5f;
        const float top = r0[k0][q] * (1.f - wx) + r0[k0 + 1][q] * wx;
        const float bot = r1[k0][q] * (1.f - wx) + r1[k0 + 1][q] * wx;
        o[r] = top * (1.f - wy) + bot * wy;
      }
      float* out = (float*)d.out + (((int64_t)b * d.C + c) * d.Hout + oy) * d.Wout + S * j;
      if (S == 4) *(float4*)out = make_float4(o[0], o[1], o[2], o[3]);
      else *(float2*)out = make_float2(o[0], o[1]);
    }
  }
}

// backward as a GATHER (deterministic, no atomics): `in` = dout (NHWC of Hout x Wout, or NCHW fp32 when out_nchw),
// `out` = din fp32 NHWC, accumulated (+=).  An input pixel i receives from the output pixels o whose source interval
// touches i: o in (  (i - 0.5)/s - 0.5 ,  (i + 1.5)/s - 0.5 ),  s = in/out  (plus the clamped borders).
MTT_DEV void gather_range(int i, int in, int out, int& lo, int& hi) {
  const float inv = (float)out / (float)in;
  float flo = ((float)i - 0.5f) * inv - 0.5f, fhi = ((float)i + 1.5f) * inv - 0.5f;
  lo = (int)floorf(flo) - 1; hi = (int)ceilf(fhi) + 1;
  if (i == 0) lo = 0;                    // clamped sources (src < 0 -> 0) all land on row 0
  if (lo < 0) lo = 0;
  if (hi > out - 1) hi = out - 1;
}
MTT_DEV float tap_weight(int o, int i, int in, int out) {
  int i0, i1; float w1;
  src_index(o, in, out, i0, i1, w1);
  float w = 0.f;
  if (i0 == i) w += 1.f - w1;
  if (i1 == i) w += w1;
  return w;
}

// Integer scale S, NCHW gradient planes: input pixel i gathers the 2 S outputs S i - S/2 .. S i + 3 S/2 - 1 of each axis with the tent
// weights (t + 1/2) / S and their mirror (constants, no per-tap weight evaluation); a border pixel also holds the share of its replicated
// neighbour (align_corners = False clamps the source index) and outputs beyond the map do not exist.  One lane = (channel, input column).
template <int S>
MTT_DEV void int_scale_weights(int i, int in, float (&w)[2 * S]) {
#pragma unroll
  for (int t = 0; t < 2 * S; ++t) w[t] = t < S ? (t + 0.5f) / S : 2.f - (t + 0.5f) / S;
  if (i == 0) {
#pragma unroll
    for (int t = 0; t < S; ++t) w[t] = t < S / 2 ? 0.f : 1.f;
  }
  if (i == in - 1) {
#pragma unroll
    for (int t = S; t < 2 * S; ++t) w[t] = t < S + S / 2 ? 1.f : 0.f;
  }
}
template <int S>
__global__ __launch_bounds__(256) void bilinear_bwd_nchw_int_kernel(const mtt_resize_desc d) {
  float* din = (float*)d.out;
  const int iy = blockIdx.y, b = blockIdx.z;
  float wy[2 * S];
  int_scale_weights<S>(iy, d.Hin, wy);
  const unsigned total = (unsigned)d.C * (unsigned)d.Win;
  for (unsigned t = blockIdx.x * 256u + threadIdx.x; t < total; t += gridDim.x * 256u) {
    const unsigned c = t / (unsigned)d.Win, ix = t - c * (unsigned)d.Win;
    float wx[2 * S];
    int_scale_weights<S>((int)ix, d.Win, wx);
    const float* g = (const float*)d.in + ((int64_t)b * d.C + c) * d.Hout * d.Wout;
    const int ox0 = S * (int)ix - S / 2, oy0 = S * iy - S / 2;
    float acc = 0.f;
#pragma unroll
    for (int u = 0; u < 2 * S; ++u) {
      const int oy = oy0 + u;
      if (oy < 0 || oy >= d.Hout) continue;                      // their weights are zero (border rows)
      const float* row = g + (int64_t)oy * d.Wout;
      float rowacc = 0.f;
      if (S == 4) {
#pragma unroll
        for (int v = 0; v < 2 * S; v += 2) {                     // S = 4: ox0 = 4 ix - 2 is even -> the pairs are 8-byte aligned and never straddle the edge
          const int ox = ox0 + v;
          float2 q = make_float2(0.f, 0.f);
          if (ox >= 0 && ox + 1 < d.Wout) q = *(const float2*)(row + ox);
          rowacc += wx[v] * q.x + wx[v + 1] * q.y;
        }
      } else {
#pragma unroll
        for (int v = 0; v < 2 * S; ++v) {
          const int ox = ox0 + v;
          if (ox >= 0 && ox < d.Wout) rowacc += wx[v] * row[ox];
        }
      }
      acc += wy[u] * rowacc;
    }
    din[(((int64_t)b * d.Hin + iy) * d.Win + ix) * d.ld_in + c] += acc;
  }
}

// grid (column blocks, input row, batch)
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const mtt_resize_desc d) {
  float* din = (float*)d.out;
  const int iy = blockIdx.y, b = blockIdx.z;
  int ylo, yhi;
  gather_range(iy, d.Hin, d.Hout, ylo, yhi);
  if (d.out_nchw) {
    // one thread per (channel, input column): reads a small window of the NCHW gradient plane
    const unsigned total = (unsigned)d.C * (unsigned)d.Win;
    for (unsigned t = blockIdx.x * 256u + threadIdx.x; t < total; t += gridDim.x * 256u) {
      const unsigned c = t / (unsigned)d.Win, ix = t - c * (unsigned)d.Win;
      int xlo, xhi;
      gather_range((int)ix, d.Win, d.Wout, xlo, xhi);
      const float* g = (const float*)d.in + ((int64_t)b * d.C + c) * d.Hout * d.Wout;
      float acc = 0.f;
      for (int oy = ylo; oy <= yhi; ++oy) {
        const float wy = tap_weight(oy, iy, d.Hin, d.Hout);
        if (wy == 0.f) continue;
        float rowacc = 0.f;
        for (int ox = xlo; ox <= xhi; ++ox) rowacc += tap_weight(ox, (int)ix, d.Win, d.Wout) * g[(int64_t)oy * d.Wout + ox];
        acc += wy * rowacc;
      }
      din[(((int64_t)b * d.Hin + iy) * d.Win + ix) * d.ld_in + c] += acc;
    }
  } else {
    const unsigned C8 = (unsigned)((d.C + 7) >> 3);
    const unsigned total = (unsigned)d.Win * C8;
    for (unsigned t = blockIdx.x * 256u + threadIdx.x; t < total; t += gridDim.x * 256u) {
      const unsigned ix = t / C8, c8 = t - ix * C8;
      const int64_t pix = ((int64_t)b * d.Hin + iy) * d.Win + ix;
      int xlo, xhi;
      gather_range((int)ix, d.Win, d.Wout, xlo, xhi);
      float acc[8];
      ld8(din, pix * d.ld_in + c8 * 8, MTT_F32, acc);
      for (int oy = ylo; oy <= yhi; ++oy) {
        const float wy = tap_weight(oy, iy, d.Hin, d.Hout);
        if (wy == 0.f) continue;
        for (int ox = xlo; ox <= xhi; ++ox) {
          const float w = wy * tap_weight(ox, (int)ix, d.Win, d.Wout);
          if (w == 0.f) continue;
          float g[8];
          ld8(d.in, (((int64_t)b * d.Hout + oy) * d.Wout + ox) * d.ld_out + c8 * 8, d.in_dtype, g);
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] += w * g[j];
        }
      }
      st8(din, pix * d.ld_in + c8 * 8, MTT_F32, acc);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Column reductions over rows of a [rows, C] NHWC matrix.  MODE 0: sum & sumsq (BN statistics);
// MODE 1: sum only (bias gradient); MODE 2: BN backward sums of du and du*xhat with du = dy*act'(u).
// ------------------------------------------------------------------------------------------------
MTT_DEV float act_fwd(float u, int act) { return act == MTT_ACT_GELU ? gelu_f(u) : (act == MTT_ACT_RELU ? fmaxf(u, 0.f) : u); }
MTT_DEV float act_bwd(float u, int act) { return act == MTT_ACT_GELU ? gelu_grad_f(u) : (act == MTT_ACT_RELU ? (u > 0.f ? 1.f : 0.f) : 1.f); }

// Column sums (bias gradients), deterministic: no atomics, fixed summation order.  Stage 1, grid (row blocks, chunks of 2048 columns):
// every lane sums its rows of one 8-column group with four independent loads in flight, the row lanes of the block are combined
// through LDS in lane order, one partial row per block goes to ws [nblk][pad8(cols)].  Stage 2 sums the partials in block order.
__global__ __launch_bounds__(256) void colsum_partial_kernel(const void* src_, float* ws, int64_t rows, int cols, int64_t ld, int dtype,
                                                             int rows_per_block, int64_t src_zs) {
  __shared__ float lsm[2048];
  // blockIdx.z = map z of a batch: source z starts src_zs elements after source z - 1, its partials after those of z - 1
  const void* src = (const unsigned char*)src_ + (int64_t)blockIdx.z * src_zs * (dtype == MTT_F32 ? 4 : 2);
  ws += (int64_t)blockIdx.z * gridDim.x * (((int64_t)cols + 7) / 8 * 8);
  const int c0 = blockIdx.y * 2048;
  const int ncol = cols - c0 < 2048 ? cols - c0 : 2048;
  const int C8 = (ncol + 7) >> 3, Cp = C8 * 8;
  const int lanes = 256 / C8;                           // >= 1 row lanes
  const int c8 = threadIdx.x % C8, rl = threadIdx.x / C8;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  if (rl < lanes) {
    float a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = 0.f;
    const int64_t col = c0 + c8 * 8;
    int64_t r = r0 + rl;
    for (; r + 3 * lanes < r1; r += 4 * lanes) {
      float v0[8], v1[8], v2[8], v3[8];
      ld8(src, r * ld + col, dtype, v0);
      ld8(src, (r + lanes) * ld + col, dtype, v1);
      ld8(src, (r + 2 * lanes) * ld + col, dtype, v2);
      ld8(src, (r + 3 * lanes) * ld + col, dtype, v3);
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] += (v0[j] + v1[j]) + (v2[j] + v3[j]);
    }
    for (; r < r1; r += lanes) {
      float v[8];
      ld8(src, r * ld + col, dtype, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] += v[j];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) lsm[rl * Cp + c8 * 8 + j] = a[j];
  }
  __syncthreads();
  const int64_t colsP = ((int64_t)cols + 7) / 8 * 8;
  for (int c = threadIdx.x; c < Cp; c += 256) {
    float t = lsm[c];
    for (int l = 1; l < lanes; ++l) t += lsm[l * Cp + c];
    ws[(int64_t)blockIdx.x * colsP + c0 + c] = t;
  }
}

// BatchNorm reductions, deterministic (no atomics) and centred.  Grid (row blocks, Z maps).  Each block reduces its rows to one
// partial per channel in `ws` [Z][nblk][2][Cp]; bn_final_kernel merges the partials in block order.
//   MODE 0 (statistics): the block accumulates sums of (x - shift) and (x - shift)^2 with shift = the block's first row (a value
//     within a few sigma of the mean, so nothing cancels even when |mean| >> std) and writes (mean_b, M2_b = sum (x - mean_b)^2);
//     the merge is Chan's pairwise update — the result is the centred variance nn.BatchNorm2d computes, not E[x^2] - E[x]^2.
//   MODE 2 (backward): sums of du and du * xhat with du = dy * act'(u).
constexpr int BNR_LSM = 2 * 2048;     // lanes * Cp <= 256 * 8 floats per plane
template <int MODE>
__global__ __launch_bounds__(256) void bn_reduce_kernel(const mtt_bn_desc d, int rows_per_block, float* ws) {
  __shared__ float lsm[BNR_LSM];
  const int C8 = (d.C + 7) >> 3, Cp = C8 * 8;
  const int lanes = 256 / C8 > 0 ? 256 / C8 : 1;
  const int c8 = threadIdx.x % C8, rl = threadIdx.x / C8;
  const int z = blockIdx.y;
  const int es = d.dtype == MTT_F32 ? 4 : 2;
  const int gdt = d.g_dtype ? d.g_dtype - 1 : d.dtype, ges = gdt == MTT_F32 ? 4 : 2;       // dy may be stored narrower than x (ABI 9)
  const unsigned char* xz = (const unsigned char*)d.x + (int64_t)z * d.x_zs * es;
  const unsigned char* dyz = (const unsigned char*)d.dy + (int64_t)z * d.x_zs * ges;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < d.rows ? r0 + rows_per_block : d.rows;
  float a0[8], a1[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) a0[j] = a1[j] = sh[j] = 0.f;
  if (rl < lanes) {
    if (MODE == 0) {
      ld8(xz, r0 * d.ld + c8 * 8, d.dtype, sh);
      for (int64_t r = r0 + rl; r < r1; r += lanes) {
        float v[8];
        ld8(xz, r * d.ld + c8 * 8, d.dtype, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float t = v[j] - sh[j]; a0[j] += t; a1[j] = fmaf(t, t, a1[j]); }
      }
    } else {
      float mu[8], rs[8], ga[8], be[8];
      const int64_t pz = (int64_t)z * d.p_zs + c8 * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bool ok = c8 * 8 + j < d.C;
        mu[j] = ok ? d.mean[pz + j] : 0.f; rs[j] = ok ? d.rstd[pz + j] : 0.f; ga[j] = ok ? d.gamma[pz + j] : 0.f; be[j] = ok ? d.beta[pz + j] : 0.f;
      }
      for (int64_t r = r0 + rl; r < r1; r += lanes) {
        float x[8], v[8];
        ld8(xz, r * d.ld + c8 * 8, d.dtype, x);
        ld8(dyz, r * d.ld + c8 * 8, gdt, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float xh = (x[j] - mu[j]) * rs[j];
          const float du = v[j] * act_bwd(xh * ga[j] + be[j], d.act);
          a0[j] += du; a1[j] = fmaf(du, xh, a1[j]);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { lsm[rl * Cp + c8 * 8 + j] = a0[j]; lsm[2048 + rl * Cp + c8 * 8 + j] = a1[j]; }
  }
  __syncthreads();
  float* out = ws + ((int64_t)z * gridDim.x + blockIdx.x) * 2 * Cp;
  for (int c = threadIdx.x; c < Cp; c += 256) {
    float s0 = 0.f, s1 = 0.f;
    for (int l = 0; l < lanes; ++l) { s0 += lsm[l * Cp + c]; s1 += lsm[2048 + l * Cp + c]; }      // fixed order
    if (MODE == 0) {
      const float n = (float)(r1 - r0);
      const float shift = c < d.C ? ld_elem(xz, r0 * d.ld + c, d.dtype) : 0.f;
      out[c] = shift + s0 / n;                    // block mean
      out[Cp + c] = fmaxf(s1 - s0 * s0 / n, 0.f); // block M2 (about the block mean)
    } else {
      out[c] = s0; out[Cp + c] = s1;
    }
  }
}

// Merge of the per-block partials, in a fixed order: block = 32 channels x 8 partial lanes (lane l takes partials l, l+8, ...), the 8
// lanes are combined through LDS by lane 0.  Grid (ceil(C / 32), Z).
template <int MODE>
__global__ __launch_bounds__(256) void bn_final_kernel(const mtt_bn_desc d, int rows_per_block, int nblk, const float* ws) {
  __shared__ float sh[3][8][32];
  const int cl = threadIdx.x & 31, pl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  const int z = blockIdx.y;
  const int Cp = ((d.C + 7) >> 3) * 8;
  const bool cok = c < d.C;
  const float* p = ws + (int64_t)z * nblk * 2 * Cp + (cok ? c : 0);
  float n = 0.f, a0 = 0.f, a1 = 0.f;                 // MODE 0: (count, mean, M2); MODE 2: (-, sum0, sum1)
  for (int b = pl; b < nblk; b += 8) {
    const float v0 = p[(int64_t)b * 2 * Cp], v1 = p[(int64_t)b * 2 * Cp + Cp];
    if (MODE == 0) {
      const int64_t r0 = (int64_t)b * rows_per_block;
      const float nb = (float)((r0 + rows_per_block < d.rows ? r0 + rows_per_block : d.rows) - r0);
      const float tot = n + nb, delta = v0 - a0;
      a0 += delta * (nb / tot);
      a1 += v1 + delta * delta * (n * nb / tot);
      n = tot;
    } else {
      a0 += v0; a1 += v1;
    }
  }
  sh[0][pl][cl] = n; sh[1][pl][cl] = a0; sh[2][pl][cl] = a1;
  __syncthreads();
  if (pl != 0 || !cok) return;
  for (int l = 1; l < 8; ++l) {
    const float nb = sh[0][l][cl], v0 = sh[1][l][cl], v1 = sh[2][l][cl];
    if (MODE == 0) {
      if (nb > 0.f) {
        const float tot = n + nb, delta = v0 - a0;
        a0 += delta * (nb / tot);
        a1 += v1 + delta * delta * (n * nb / tot);
        n = tot;
      }
    } else {
      a0 += v0; a1 += v1;
    }
  }
  float* const q0 = MODE == 0 ? d.mean_out : d.dsum;
  float* const q1 = MODE == 0 ? d.m2_out : d.dsumxh;
  q0[(int64_t)z * d.p_zs + c] = a0;
  q1[(int64_t)z * d.p_zs + c] = a1;
}

// y = act((x - mean) * rstd * gamma + beta); channels >= C are written as zeros (padding).
// Block = a range of rows; thread = one fixed 8-channel chunk (per-channel parameters live in registers, no
// division in the loop), 256 / C8 rows in flight per block.
template <bool BWD>
__global__ __launch_bounds__(256) void bn_rowwise_kernel(const mtt_bn_desc d, int rows_per_block) {
  const int C8 = (int)(d.ld >> 3);                         // the whole pitch: channels C .. ld-1 are written as zeros
  const int lanes = C8 >= 256 ? 1 : 256 / C8;
  const int c8_0 = C8 >= 256 ? threadIdx.x : threadIdx.x % C8;
  const int rl = C8 >= 256 ? 0 : threadIdx.x / C8;
  if (rl >= lanes) return;
  const int z = blockIdx.y;
  const int es = d.dtype == MTT_F32 ? 4 : 2;
  const int gdt = (BWD && d.g_dtype) ? d.g_dtype - 1 : d.dtype, ges = gdt == MTT_F32 ? 4 : 2;    // BWD: dy / dx may be stored narrower than x (ABI 9)
  const int64_t zoff = (int64_t)z * d.x_zs * es, goff = (int64_t)z * d.x_zs * ges, pz = (int64_t)z * d.p_zs;
  const unsigned char* xz = (const unsigned char*)d.x + zoff;
  const unsigned char* dyz = (const unsigned char*)d.dy + goff;
  unsigned char* oz = (unsigned char*)(BWD ? d.dx : d.y) + (BWD ? goff : zoff);
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < d.rows ? r0 + rows_per_block : d.rows;
  const float invn = 1.0f / (float)d.rows;
  for (int c8 = c8_0; c8 < C8; c8 += 256) {
    float mu[8], rs[8], ga[8], be[8], s0[8], s1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = c8 * 8 + j;
      const bool ok = c < d.C;
      mu[j] = ok ? d.mean[pz + c] : 0.f; rs[j] = ok ? d.rstd[pz + c] : 0.f; ga[j] = ok ? d.gamma[pz + c] : 0.f; be[j] = ok ? d.beta[pz + c] : 0.f;
      if (BWD) { s0[j] = ok ? d.dsum[pz + c] * invn : 0.f; s1[j] = ok ? d.dsumxh[pz + c] * invn : 0.f; }
    }
    const bool tail = c8 * 8 + 8 > d.C;
    // one row of 8 channels per lane and iteration.  (Four rows per iteration — their loads in flight together — measured SLOWER: 104 instead
    // of 56 VGPRs halves the waves in flight, 1 557 vs 1 157 us on the head maps: profiles/r03_train_ns6_b63_g.txt vs _d.txt.)
    for (int64_t r = r0 + rl; r < r1; r += lanes) {
      float x[8], o[8];
      ld8(xz, r * d.ld + c8 * 8, d.dtype, x);
      if (BWD) {
        float g[8];
        ld8(dyz, r * d.ld + c8 * 8, gdt, g);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float xh = (x[j] - mu[j]) * rs[j];
          const float du = g[j] * act_bwd(xh * ga[j] + be[j], d.act);
          o[j] = ga[j] * rs[j] * (du - s0[j] - xh * s1[j]);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = act_fwd((x[j] - mu[j]) * rs[j] * ga[j] + be[j], d.act);
      }
      if (tail) {
#pragma unroll
        for (int j = 0; j < 8; ++j) if (c8 * 8 + j >= d.C) o[j] = 0.f;
      }
      st8(oz, r * d.ld + c8 * 8, BWD ? gdt : d.dtype, o);
    }
  }
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cast2d_kernel(const void* src, void* dst, int64_t rows, int64_t cols, int64_t lds_, int64_t ldd,
                                                     int sdt, int ddt, int zero_pad) {
  const int64_t wcols = zero_pad ? ldd : cols;
  const int64_t total = rows * wcols;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int64_t r = t / wcols, c = t % wcols;
    st_elem(dst, r * ldd + c, ddt, c < cols ? ld_elem(src, r * lds_ + c, sdt) : 0.f);
  }
}

// multi-segment strided copy / cast (mtt_segcopy): one workgroup = one chunk of SEG_CHUNK logical elements of one segment
constexpr int SEG_CHUNK = 16384;
MTT_DEV void seg_store(int ddt, void* dst, void* dst_lo, int64_t o, float v) {
  if (ddt == MTT_F32) ((float*)dst)[o] = v;
  else {
    const bf16_t h = f2bf(v);
    ((bf16_t*)dst)[o] = h;
    if (ddt == MTT_SPLIT) ((bf16_t*)dst_lo)[o] = f2bf(v - bf2f(h));
  }
}
__global__ __launch_bounds__(256) void segcopy_kernel(const mtt_segcopy_desc d) {
  const int64_t* g = d.table + (int64_t)d.chunk_seg[blockIdx.x] * MTT_SEG_WORDS;
  const char* src = (const char*)(g[0] + d.src_base);
  char* dst = (char*)(g[1] + d.dst_base);
  char* dst_lo = g[2] ? (char*)(g[2] + d.dst_base) : nullptr;
  const int64_t total = g[3], n1 = g[4], n2 = g[5];
  const int64_t s0 = g[6], s1 = g[7], s2 = g[8], d0 = g[9], d1 = g[10], d2 = g[11];
  const int sdt = (int)g[12], ddt = (int)g[13];
  const int64_t e0 = d.chunk_off[blockIdx.x];
  const int64_t e1 = e0 + SEG_CHUNK < total ? e0 + SEG_CHUNK : total;
  if (g[15]) {                                        // transposing segment: 64 x 64 tiles of (i1, i2) through LDS; s1 == 1, d2 == 1
    __shared__ float tile[64][65];
    const int64_t t1 = (n1 + 63) >> 6, t2 = (n2 + 63) >> 6;
    const int64_t tile0 = e0 >> 12, tile1 = tile0 + SEG_CHUNK / 4096 < total >> 12 ? tile0 + SEG_CHUNK / 4096 : total >> 12;
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    for (int64_t t = tile0; t < tile1; ++t) {
      const int64_t j2 = t % t2, r = t / t2, j1 = r % t1, i0 = r / t1;
      __syncthreads();
      for (int k = ly; k < 64; k += 4) {                 // row k of the tile = i2, lanes along i1 (contiguous in the source)
        const int64_t i1 = j1 * 64 + lx, i2 = j2 * 64 + k;
        if (i1 < n1 && i2 < n2) tile[k][lx] = ld_elem(src, i0 * s0 + i1 + i2 * s2, sdt);
      }
      __syncthreads();
      for (int k = ly; k < 64; k += 4) {                 // lanes along i2 (contiguous in the destination)
        const int64_t i1 = j1 * 64 + k, i2 = j2 * 64 + lx;
        if (i1 < n1 && i2 < n2) seg_store(ddt, dst, dst_lo, i0 * d0 + i1 * d1 + i2, tile[lx][k]);
      }
    }
    return;
  }
  if (g[14]) {                                        // four consecutive elements of a row per lane
    for (int64_t e = e0 + threadIdx.x * 4; e < e1; e += 256 * 4) {
      const int64_t i2 = e % n2, r = e / n2, i1 = r % n1, i0 = r / n1;
      const int64_t so = i0 * s0 + i1 * s1 + i2, o = i0 * d0 + i1 * d1 + i2;
      float v[4];
      if (sdt == MTT_F32) { const float4 q = *(const float4*)((const float*)src + so); v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; }
      else { const uint2 q = *(const uint2*)((const bf16_t*)src + so); v[0] = lo_of(q.x); v[1] = hi_of(q.x); v[2] = lo_of(q.y); v[3] = hi_of(q.y); }
      if (ddt == MTT_F32) *(float4*)((float*)dst + o) = make_float4(v[0], v[1], v[2], v[3]);
      else {
        const uint2 h = make_uint2(pack2(v[0], v[1]), pack2(v[2], v[3]));
        *(uint2*)((bf16_t*)dst + o) = h;
        if (ddt == MTT_SPLIT)
          *(uint2*)((bf16_t*)dst_lo + o) = make_uint2(pack2(v[0] - lo_of(h.x), v[1] - hi_of(h.x)), pack2(v[2] - lo_of(h.y), v[3] - hi_of(h.y)));
      }
    }
    return;
  }
  for (int64_t e = e0 + threadIdx.x; e < e1; e += 256) {
    const int64_t i2 = e % n2, r = e / n2, i1 = r % n1, i0 = r / n1;
    seg_store(ddt, dst, dst_lo, i0 * d0 + i1 * d1 + i2 * d2, ld_elem(src, i0 * s0 + i1 * s1 + i2 * s2, sdt));
  }
}

// fp32 -> split planes, 8 columns per lane (cols padded to ldd with zeros)
__global__ __launch_bounds__(256) void split_cast_kernel(const float* src, bf16_t* hi, bf16_t* lo, int64_t rows, int64_t cols, int64_t lds_, int64_t ldd) {
  const int64_t c8n = ldd >> 3, total = rows * c8n;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int64_t r = t / c8n, c = (t - r * c8n) * 8;
    float v[8];
    if (c + 8 <= cols && (lds_ & 3) == 0) ld8(src, r * lds_ + c, MTT_F32, v);
    else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = c + j < cols ? src[r * lds_ + c + j] : 0.f;
    }
    const u32x4 h = (u32x4){pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
    *(u32x4*)(hi + r * ldd + c) = h;
    *(u32x4*)(lo + r * ldd + c) = (u32x4){pack2(v[0] - lo_of(h.x), v[1] - hi_of(h.x)), pack2(v[2] - lo_of(h.y), v[3] - hi_of(h.y)),
                                         pack2(v[4] - lo_of(h.z), v[5] - hi_of(h.z)), pack2(v[6] - lo_of(h.w), v[7] - hi_of(h.w))};
  }
}

__global__ __launch_bounds__(256) void add_rows_kernel(const void* src, float* dst, int64_t rows, int cols, int64_t lds_, int64_t ldd,
                                                       int sdt, float alpha) {
  const int64_t total = rows * cols;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int64_t r = t / cols, c = t % cols;
    dst[r * ldd + c] += alpha * ld_elem(src, r * lds_ + c, sdt);
  }
}


// ------------------------------------------------------------------------------------------------
// Backward of the task-feature modulation.  grid (C/8/32 column groups, pixel splits, B*nwin); block =
// 32 column chunks x 8 pixel lanes (same shape as chanlogit_kernel).
//   dx[b,p,c]               = sum_t dout[2t]*(1+a) + dout[2t+1]*(1+bw)                          (written)
//   drawlog[b,head,t,T+p]   = sum_{c in head} dout[2t][b,p,c] * x[b,p,c]         (unique owner: 8 lanes)
//   drawchan[b,t,win,c]    += sum_{p in win} dout[2t+1][b,p,c] * x[b,p,c]        (LDS reduce + 1 atomic)
// ------------------------------------------------------------------------------------------------
// NT = tasks of this launch (compile time: exact accumulator count and no per-task predicates — the run-time form with CL_MAXT slots spilled)
template <int NT>
__global__ __launch_bounds__(256) void modulate_bwd_kernel(const mtt_modulate_desc d, const void* dout, float* dx,
                                                           float* drawlog, float* part, int64_t plane_elems, int tbase) {
  const int nwin = d.nh * d.nw, wh = d.h / d.nh, ww = d.w / d.nw, P = wh * ww, hw = d.h * d.w;
  const int hg = d.hg > 0 ? d.hg : 64;             // 64 or 32 channels per head (host-checked)
  const int nH = d.C / hg;
  const int b = blockIdx.z / nwin, win = blockIdx.z % nwin;
  const int wy = win / d.nw, wx = win % d.nw;
  const int cl = threadIdx.x & 31, plane = threadIdx.x >> 5;
  const int cchunk = blockIdx.x * 32 + cl;
  const bool cok = cchunk * 8 < d.C;
  constexpr int nT = NT;
  const int64_t planeElems = (int64_t)d.B * hw * d.C;
  float acc[NT][8];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[t][j] = 0.f;
  const int per = (P + gridDim.y - 1) / gridDim.y;
  const int p0 = blockIdx.y * per, p1 = p0 + per < P ? p0 + per : P;
  const int head = (cchunk * 8) / hg;
  const int lanes_per_head = hg >> 3;              // 8 (or 4) consecutive lanes hold the chunks of one head
  for (int pi0 = p0; pi0 < p1; pi0 += 8) {
    const int pi = pi0 + plane;
    const bool pok = pi < p1 && cok;
    const int pic = pi < p1 ? pi : p1 - 1;
    const int y = wy * wh + pic / ww, x = wx * ww + pic % ww;
    const int pix = y * d.w + x;
    const int64_t tok = (int64_t)b * hw + pix;
    float xv[8], dxa[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { xv[j] = 0.f; dxa[j] = 0.f; }
    if (pok) ld8(d.x, (int64_t)b * d.x_bs + (int64_t)pix * d.x_ld + cchunk * 8, MTT_F32, xv);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      {
        const int tk = tbase + t;
        float gs[8], gc[8], bw[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { gs[j] = gc[j] = bw[j] = 0.f; }
        float a = 0.f;
        if (pok) {
          ld8(dout, (int64_t)(2 * tk) * planeElems + tok * d.C + cchunk * 8, d.out_dtype, gs);
          ld8(dout, (int64_t)(2 * tk + 1) * planeElems + tok * d.C + cchunk * 8, d.out_dtype, gc);
          ld8(d.rawchan, (((int64_t)b * d.T + tk) * nwin + win) * d.C + cchunk * 8, MTT_F32, bw);
          a = d.rawlog[(((int64_t)b * nH + head) * d.T + tk) * d.N + d.T + pix];
        }
        float dl = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          dxa[j] += gs[j] * (1.0f + a) + gc[j] * (1.0f + bw[j]);
          dl += gs[j] * xv[j];
          acc[t][j] += gc[j] * xv[j];
        }
        // the chunks of one head are consecutive lanes
        dl += __shfl_xor(dl, 1, 64); dl += __shfl_xor(dl, 2, 64);
        if (lanes_per_head == 8) dl += __shfl_xor(dl, 4, 64);
        if (pok && (cl & (lanes_per_head - 1)) == 0) drawlog[(((int64_t)b * nH + head) * d.T + tk) * d.N + d.T + pix] = dl;
      }
      if (t & 1) __builtin_amdgcn_sched_barrier(0);      // two tasks' loads in flight at a time: hoisting all NT of them costs 36 VGPRs per task
    }
    if (pok && tbase == 0) {                             // the first task group WRITES dx (nothing to pre-zero), later groups add to it
      st8(dx, (int64_t)b * d.x_bs + (int64_t)pix * d.x_ld + cchunk * 8, MTT_F32, dxa);
    } else if (pok) {
      float cur[8];
      const int64_t xi = (int64_t)b * d.x_bs + (int64_t)pix * d.x_ld + cchunk * 8;
      ld8(dx, xi, MTT_F32, cur);
#pragma unroll
      for (int j = 0; j < 8; ++j) cur[j] += dxa[j];
      st8(dx, xi, MTT_F32, cur);
    }
  }
  __shared__ float red[8][32][8];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int j = 0; j < 8; ++j) red[plane][cl][j] = acc[t][j];
    __syncthreads();
    if (plane == 0 && cok) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) s += red[q][cl][j];
        part[(int64_t)blockIdx.y * plane_elems + (((int64_t)b * d.T + tbase + t) * nwin + win) * d.C + cchunk * 8 + j] = s;
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// Backward of the channel-attention logits: one wave per patch token.
//   dq[b,t,p]      = sum_c drawchan[b,t,win(p),c] * xn[b,T+p,c]
//   dxn[b,T+p,c]  += sum_t drawchan[b,t,win(p),c] * q[b,t,p]           (fp32 accumulate)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void chanlogit_bwd_kernel(const mtt_chanlogit_desc d, const float* drawchan, void* dq, int dq_dtype,
                                                            float* dxn) {
  const int lane = threadIdx.x & 63;
  const int hw = d.h * d.w, nwin = d.nh * d.nw, wh = d.h / d.nh, ww = d.w / d.nw;
  const int64_t tokid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tokid >= (int64_t)d.B * hw) return;
  const int b = (int)(tokid / hw), p = (int)(tokid % hw);
  const int y = p / d.w, x = p % d.w;
  const int win = (y / wh) * d.nw + (x / ww);
  const int64_t xrow = ((int64_t)b * d.N + d.T + p) * d.C;
  const int C8 = d.C >> 3;
  float qv[CL_MAXT * 2], sq[CL_MAXT * 2];
  const int nT = d.T < CL_MAXT * 2 ? d.T : CL_MAXT * 2;
#pragma unroll
  for (int t = 0; t < CL_MAXT * 2; ++t) {
    sq[t] = 0.f;
    qv[t] = t < nT ? ld_elem(d.q, ((int64_t)b * d.T + t) * d.ldq + p, d.dtype) : 0.f;
  }
  for (int c8 = lane; c8 < C8; c8 += 64) {
    float xv[8], da[8];
    ld8(d.xn, xrow + c8 * 8, d.dtype, xv);
    ld8(dxn, xrow + c8 * 8, MTT_F32, da);
#pragma unroll
    for (int t = 0; t < CL_MAXT * 2; ++t) {
      if (t < nT) {
        float g[8];
        ld8(drawchan, (((int64_t)b * d.T + t) * nwin + win) * d.C + c8 * 8, MTT_F32, g);
#pragma unroll
        for (int j = 0; j < 8; ++j) { sq[t] += g[j] * xv[j]; da[j] += g[j] * qv[t]; }
      }
    }
    st8(dxn, xrow + c8 * 8, MTT_F32, da);
  }
#pragma unroll
  for (int t = 0; t < CL_MAXT * 2; ++t) {
    if (t < nT) {
      const float s = wave_sum(sq[t]);
      if (lane == 0) st_elem(dq, ((int64_t)b * d.T + t) * d.ldq + p, dq_dtype, s);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// dwmix[b,t,s] += sum_{rows in b, c} dout[t][row,c] * fea[s][row,c]
// Token-grouped form (T <= 8, window width % 4 == 0): a wave takes FOUR consecutive pixels of one window row, so the T gradient rows
// drawchan[b, t, win, :] it multiplies with — 4 KiB each, served by L2 — are loaded once per group instead of once per pixel (the
// one-pixel form above moved 24 KiB of L2 traffic per pixel for 10 KiB of HBM traffic and ran at 1.0-1.6 TB/s).
__global__ __launch_bounds__(256) void chanlogit_bwd_tok4_kernel(const mtt_chanlogit_desc d, const float* drawchan, void* dq, int dq_dtype,
                                                                 float* dxn) {
  constexpr int TOK = 4;
  const int lane = threadIdx.x & 63;
  const int hw = d.h * d.w, nwin = d.nh * d.nw, wh = d.h / d.nh, ww = d.w / d.nw;
  const int64_t gid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (gid >= (int64_t)d.B * hw / TOK) return;
  const int b = (int)(gid / (hw / TOK)), p0 = (int)(gid % (hw / TOK)) * TOK;
  const int y = p0 / d.w, x = p0 % d.w;
  const int win = (y / wh) * d.nw + (x / ww);
  const int64_t xrow = ((int64_t)b * d.N + d.T + p0) * d.C;
  const int C8 = d.C >> 3;
  float qv[CL_MAXT][TOK], sq[CL_MAXT][TOK];
#pragma unroll
  for (int t = 0; t < CL_MAXT; ++t)
#pragma unroll
    for (int k = 0; k < TOK; ++k) {
      sq[t][k] = 0.f;
      qv[t][k] = t < d.T ? ld_elem(d.q, ((int64_t)b * d.T + t) * d.ldq + p0 + k, d.dtype) : 0.f;
    }
#pragma unroll 1
  for (int c8 = lane; c8 < C8; c8 += 64) {
    float xv[TOK][8], da[TOK][8];
#pragma unroll
    for (int k = 0; k < TOK; ++k) {
      ld8(d.xn, xrow + (int64_t)k * d.C + c8 * 8, d.dtype, xv[k]);
      ld8(dxn, xrow + (int64_t)k * d.C + c8 * 8, MTT_F32, da[k]);
    }
#pragma unroll
    for (int t = 0; t < CL_MAXT; ++t)
      if (t < d.T) {
        float g[8];
        ld8(drawchan, (((int64_t)b * d.T + t) * nwin + win) * d.C + c8 * 8, MTT_F32, g);
#pragma unroll
        for (int k = 0; k < TOK; ++k)
#pragma unroll
          for (int j = 0; j < 8; ++j) { sq[t][k] += g[j] * xv[k][j]; da[k][j] += g[j] * qv[t][k]; }
      }
#pragma unroll
    for (int k = 0; k < TOK; ++k) st8(dxn, xrow + (int64_t)k * d.C + c8 * 8, MTT_F32, da[k]);
  }
#pragma unroll
  for (int t = 0; t < CL_MAXT; ++t)
    if (t < d.T) {
#pragma unroll
      for (int k = 0; k < TOK; ++k) {
        const float v = wave_sum(sq[t][k]);
        if (lane == 0) st_elem(dq, ((int64_t)b * d.T + t) * d.ldq + p0 + k, dq_dtype, v);
      }
    }
}
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ctr_dw_kernel(const mtt_ctr_desc d, const float* dout, float* part, int rows_per_block) {
  __shared__ float red[4][CTR_MAXT * CTR_MAXT];
  const int C8 = (d.C + 7) >> 3;
  const int b = blockIdx.y;
  const int64_t rows = (int64_t)d.B * d.rows_per_b, plane = rows * d.ld;
  const int64_t r0 = (int64_t)b * d.rows_per_b + (int64_t)blockIdx.x * rows_per_block;
  const int64_t rend = (int64_t)(b + 1) * d.rows_per_b;
  const int64_t r1 = r0 + rows_per_block < rend ? r0 + rows_per_block : rend;
  float acc[CTR_MAXT][CTR_MAXT];
#pragma unroll
  for (int t = 0; t < CTR_MAXT; ++t)
#pragma unroll
    for (int s = 0; s < CTR_MAXT; ++s) acc[t][s] = 0.f;
  const int64_t total = (r1 - r0) * C8;
  for (int64_t i = threadIdx.x; i < total; i += 256) {
    const int c8 = (int)(i % C8);
    const int64_t row = r0 + i / C8;
    float f[CTR_MAXT][8];
#pragma unroll
    for (int s = 0; s < CTR_MAXT; ++s)
      if (s < d.T) ld8(d.fea, (int64_t)s * plane + row * d.ld + c8 * 8, d.fea_dtype, f[s]);
#pragma unroll
    for (int t = 0; t < CTR_MAXT; ++t)
      if (t < d.T) {
        float g[8];
        ld8(dout, (int64_t)t * plane + row * d.ld + c8 * 8, MTT_F32, g);
#pragma unroll
        for (int s = 0; s < CTR_MAXT; ++s)
          if (s < d.T) {
            float v = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) v += g[j] * f[s][j];
            acc[t][s] += v;
          }
      }
  }
#pragma unroll
  for (int t = 0; t < CTR_MAXT; ++t)
#pragma unroll
    for (int s = 0; s < CTR_MAXT; ++s)
      if (t < d.T && s < d.T) {
        const float v = wave_sum(acc[t][s]);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][t * CTR_MAXT + s] = v;
      }
  __syncthreads();
  // this block's partial -> plane blockIdx.x of the workspace (summed in block order by mtt_reduce_few_kernel: no atomics)
  const int64_t n = (int64_t)d.B * d.T * d.T;
  for (int i = threadIdx.x; i < d.T * d.T; i += 256) {
    const int t = i / d.T, s = i % d.T, j = t * CTR_MAXT + s;
    part[(int64_t)blockIdx.x * n + ((int64_t)b * d.T + t) * d.T + s] = (red[0][j] + red[1][j]) + (red[2][j] + red[3][j]);
  }
}

// vectorised form (8-column chunks, 32-bit index math): cols, both pitches multiples of 8 and rows * cols / 8 < 2^31
__global__ __launch_bounds__(256) void rowscale_cast_vec_kernel(const void* src, void* dst, int rows, int C8, int64_t lds_, int64_t ldd, int sdt,
                                                                int ddt, const float* rowscale, int mb, int n_prompt) {
  const unsigned total = (unsigned)rows * (unsigned)C8;
  for (unsigned t = blockIdx.x * 256u + threadIdx.x; t < total; t += gridDim.x * 256u) {
    const unsigned r = t / (unsigned)C8, c8 = t - r * (unsigned)C8;
    float rs = 1.0f;
    if (rowscale) {
      const unsigned q = mb > 0 ? r / (unsigned)mb : 0u, rem = mb > 0 ? r - q * (unsigned)mb : r;
      rs = rowscale[q * 2 + (rem >= (unsigned)n_prompt ? 1 : 0)];
    }
    float v[8];
    ld8(src, (int64_t)r * lds_ + c8 * 8, sdt, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] *= rs;
    st8(dst, (int64_t)r * ldd + c8 * 8, ddt, v);
  }
}

// rowscale cast + column sums of the values as stored (mtt_rowscale_cast_colsum): grid (row blocks, 256-column panels); a thread owns one
// 8-column chunk of the panel and every 8th row of the block, the 8 row groups are added in index order through LDS and the block's
// partial row goes to ws [nblk][pad8(cols)] (second stage: mtt_colsum_final_kernel).  No atomics: the bias gradient is deterministic.
__global__ __launch_bounds__(256) void rowscale_cast_colsum_kernel(const void* src, void* dst, int rows, int C8, int64_t lds_, int64_t ldd, int sdt,
                                                                   int ddt, const float* rowscale, int mb, int n_prompt, float* part, int rpb) {
  __shared__ float red[8][256];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c8 = blockIdx.y * 32 + tx;
  const int r0 = blockIdx.x * rpb, r1 = r0 + rpb < rows ? r0 + rpb : rows;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c8 < C8) {
    for (int r = r0 + ty; r < r1; r += 8) {
      float rs = 1.0f;
      if (rowscale) {
        const unsigned q = mb > 0 ? (unsigned)r / (unsigned)mb : 0u, rem = mb > 0 ? (unsigned)r - q * (unsigned)mb : (unsigned)r;
        rs = rowscale[q * 2 + (rem >= (unsigned)n_prompt ? 1 : 0)];
      }
      float v[8];
      ld8(src, (int64_t)r * lds_ + c8 * 8, sdt, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] *= rs;
      st8(dst, (int64_t)r * ldd + c8 * 8, ddt, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += ddt == MTT_BF16 ? bf2f(f2bf(v[j])) : v[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[ty][tx * 8 + j] = acc[j];
  __syncthreads();
  const int col = blockIdx.y * 256 + threadIdx.x;
  if (col < C8 * 8) {
    float t = red[0][threadIdx.x];
#pragma unroll
    for (int g = 1; g < 8; ++g) t += red[g][threadIdx.x];
    part[(int64_t)blockIdx.x * (C8 * 8) + col] = t;
  }
}

// dst[r, :] = rowscale(r) * src[r, :]  (dtype cast; DropPath scale of the branch gradient)
__global__ __launch_bounds__(256) void rowscale_cast_kernel(const void* src, void* dst, int64_t rows, int cols, int64_t lds_, int64_t ldd,
                                                            int sdt, int ddt, const float* rowscale, int mb, int n_prompt) {
  const int64_t total = rows * cols;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int64_t r = t / cols, c = t % cols;
    float rs = 1.0f;
    if (rowscale) {
      const int64_t q = mb > 0 ? r / mb : 0, rem = mb > 0 ? r % mb : r;
      rs = rowscale[q * 2 + (rem >= n_prompt ? 1 : 0)];
    }
    st_elem(dst, r * ldd + c, ddt, rs * ld_elem(src, r * lds_ + c, sdt));
  }
}


int grid_for(int64_t work_items) {
  int64_t g = (work_items + 255) / 256;
  if (g > 256 * 8) g = 256 * 8;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

#define S_ ((hipStream_t)stream)
#define LAUNCH_OK() ((int)hipGetLastError())

extern "C" int mtt_layernorm_fwd(const mtt_ln_desc* d, void* stream) {
  if (!d || !d->x || !d->y || !d->gamma || !d->beta || d->rows <= 0 || d->C <= 0) return MTT_E_BADARG;
  if ((d->C % 4) || (d->ldx % 4) || (d->ldy % 4)) return MTT_E_ALIGN;
  if (d->y_dtype == MTT_SPLIT && !d->y_lo) return MTT_E_BADARG;
  if (d->y32 && (d->ldy32 % 4)) return MTT_E_ALIGN;
  if (d->C <= 1024 && !(((uintptr_t)d->x | (uintptr_t)d->gamma | (uintptr_t)d->beta) & 15)) {
    int64_t nb = (d->rows + 3) / 4; if (nb > 256 * 16) nb = 256 * 16;           // a wave walks rows with a grid stride
    hipLaunchKernelGGL(ln_fwd_reg_kernel, dim3((unsigned)nb), dim3(256), 0, S_, *d);
  } else {
    hipLaunchKernelGGL(ln_fwd_kernel, dim3((unsigned)((d->rows + 3) / 4)), dim3(256), 0, S_, *d);
  }
  return LAUNCH_OK();
}

static void ln_bwd_cfg(const mtt_ln_desc* d, bool& fused, int& nblk, int& rpb) {
  fused = d->dx && d->dgamma && d->dbeta && d->C <= 1024 && (d->ldy % 4) == 0 &&
          !(((uintptr_t)d->x | (uintptr_t)d->dx | (uintptr_t)d->dy | (uintptr_t)d->dx_in) & 15);
  int64_t nb = fused ? (d->rows + 63) / 64 : (d->rows + 31) / 32;
  const int64_t cap = fused ? 2048 : 1024;
  if (nb > cap) nb = cap;
  if (nb < 1) nb = 1;
  rpb = (int)((d->rows + nb - 1) / nb);
  nblk = (int)((d->rows + rpb - 1) / rpb);
}
extern "C" size_t mtt_layernorm_bwd_ws_floats(int64_t rows, int32_t C) {
  if (rows <= 0 || C <= 0) return 0;
  return (size_t)2048 * 2 * (size_t)C;                   // upper bound over both kernel choices (<= 2048 row blocks)
}
extern "C" int mtt_layernorm_bwd(const mtt_ln_desc* d, void* stream) {
  if (!d || !d->x || !d->dy || !d->gamma || !d->mean || !d->rstd || d->rows <= 0 || d->C <= 0) return MTT_E_BADARG;
  if (d->C > 8192 || (d->C % 4) || (d->ldx % 4)) return MTT_E_UNSUPPORTED;
  if (d->dgamma && (!d->dbeta || !d->ws)) return MTT_E_BADARG;
  bool fused; int nblk, rpb;
  ln_bwd_cfg(d, fused, nblk, rpb);
  if (fused) {
    hipLaunchKernelGGL(ln_bwd_fused_kernel, dim3((unsigned)nblk), dim3(256), 0, S_, *d, rpb);
  } else {
    if (d->dx) hipLaunchKernelGGL(ln_bwd_dx_kernel, dim3((unsigned)((d->rows + 3) / 4)), dim3(256), 0, S_, *d);
    if (d->dgamma) {
      const int C4 = d->C / 4, lanes = C4 < 256 ? 256 / C4 : 1;
      hipLaunchKernelGGL(ln_bwd_dgb_kernel, dim3((unsigned)nblk), dim3(256), (size_t)lanes * 2 * d->C * sizeof(float), S_, *d, rpb);
    }
  }
  if (d->dgamma)
    hipLaunchKernelGGL(ln_dgb_final_kernel, dim3((unsigned)((d->C + 31) / 32), 2), dim3(256), 0, S_, (const float*)d->ws, d->dgamma, d->dbeta, d->C, nblk);
  return LAUNCH_OK();
}

extern "C" int mtt_softmax_fwd(const mtt_softmax_desc* d, void* stream) {
  if (!d || !d->S || !d->P || d->rows <= 0 || d->cols <= 0 || d->ld < d->cols) return MTT_E_BADARG;
  if (d->ld % 8) return MTT_E_ALIGN;
  hipLaunchKernelGGL(softmax_fwd_kernel, dim3((unsigned)((d->rows + 3) / 4)), dim3(256), 0, S_, *d);
  return LAUNCH_OK();
}
extern "C" int mtt_softmax_bwd(const mtt_softmax_desc* d, void* stream) {
  if (!d || !d->P || !d->dP || !d->dS || d->rows <= 0 || d->cols <= 0 || d->ld < d->cols) return MTT_E_BADARG;
  if (d->ld % 8) return MTT_E_ALIGN;
  hipLaunchKernelGGL(softmax_bwd_kernel, dim3((unsigned)((d->rows + 3) / 4)), dim3(256), 0, S_, *d);
  return LAUNCH_OK();
}

extern "C" int mtt_patchify16(const float* img, void* cols, int B, int H, int W, int out_dtype, void* stream) {
  if (!img || !cols || B <= 0 || (H % 16) || (W % 16)) return MTT_E_BADARG;
  hipLaunchKernelGGL(patchify_kernel, dim3(grid_for((int64_t)B * (H / 16) * (W / 16) * 96)), dim3(256), 0, S_, img, cols, B, H, W, out_dtype);
  return LAUNCH_OK();
}

static int chanlogit_splits(const mtt_chanlogit_desc* d) {
  const int P = (d->h / d->nh) * (d->w / d->nw);
  const int base = ((d->C + 63) / 64) * d->B * d->nh * d->nw;
  // (NS-6: 1 008 workgroups before splitting; two pixel splits measured 76 us against 85 us unsplit, profiles/r03_train_ns6_b63_l.txt / _t.txt)
  int splits = (1024 + base - 1) / base; if (splits > P / 32) splits = P / 32; if (splits < 1) splits = 1; if (splits > 64) splits = 64;
  return splits;
}
static int modulate_bwd_splits(const mtt_modulate_desc* d) {
  const int P = (d->h / d->nh) * (d->w / d->nw);
  int splits = P / 64; if (splits < 1) splits = 1; if (splits > 32) splits = 32;
  return splits;
}
static int ctr_dw_blocks(const mtt_ctr_desc* d, int* rpb_out) {
  int nb = (int)((d->rows_per_b + 127) / 128); if (nb > 256) nb = 256; if (nb < 1) nb = 1;
  const int rpb = (int)((d->rows_per_b + nb - 1) / nb);
  if (rpb_out) *rpb_out = rpb;
  return (int)((d->rows_per_b + rpb - 1) / rpb);
}
// floats of caller-owned workspace the reductions of an entry point need (0: none) — see mtt_hip.h
extern "C" size_t mtt_chan_logits_ws_floats(const mtt_chanlogit_desc* d) {
  if (!d || d->nh <= 0 || d->nw <= 0) return 0;
  const int s = chanlogit_splits(d);
  return s > 1 ? (size_t)s * d->B * d->T * d->nh * d->nw * d->C : 0;
}
extern "C" size_t mtt_modulate_bwd_ws_floats(const mtt_modulate_desc* d) {
  if (!d || d->nh <= 0 || d->nw <= 0) return 0;
  return (size_t)modulate_bwd_splits(d) * d->B * d->T * d->nh * d->nw * d->C;
}
extern "C" size_t mtt_ctr_dw_ws_floats(const mtt_ctr_desc* d) {
  if (!d || d->T <= 0) return 0;
  return (size_t)ctr_dw_blocks(d, nullptr) * d->B * d->T * d->T;
}
extern "C" int mtt_chan_logits(const mtt_chanlogit_desc* d, void* stream) {
  if (!d || !d->q || !d->xn || !d->rawchan) return MTT_E_BADARG;
  if (d->nh <= 0 || d->nw <= 0 || (d->h % d->nh) || (d->w % d->nw) || (d->C % 8)) return MTT_E_BADARG;
  if (d->dtype == MTT_SPLIT && !d->xn_lo) return MTT_E_BADARG;
  const int splits = chanlogit_splits(d);
  if (splits > 1 && !d->ws) return MTT_E_BADARG;
  const int64_t n = (int64_t)d->B * d->T * d->nh * d->nw * d->C;
  dim3 grid((d->C + 63) / 64, splits, d->B * d->nh * d->nw);
  const bool px8 = ((d->w / d->nw) % 8) == 0 && (d->ldq % 8) == 0 && (d->w % 8) == 0 && !((uintptr_t)d->q & 31);
  for (int tb = 0; tb < d->T; tb += CL_MAXT) {
    if (px8) hipLaunchKernelGGL(chanlogit_px8_kernel, grid, dim3(256), 0, S_, *d, tb, d->ws, n);
    else hipLaunchKernelGGL(chanlogit_kernel, grid, dim3(256), 0, S_, *d, tb, d->ws, n);
  }
  if (splits > 1) hipLaunchKernelGGL(mtt_reduce_few_kernel, dim3(mtt_reduce_few_grid(n)), dim3(256), 0, S_, (const float*)d->ws, splits, n, d->rawchan, 0);
  return LAUNCH_OK();
}

extern "C" int mtt_modulate(const mtt_modulate_desc* d, void* stream) {
  if (!d || !d->x || !d->rawlog || !d->rawchan || !d->out || d->hg < 0 || (d->hg % 8) || (d->C % (d->hg > 0 ? d->hg : 64))) return MTT_E_BADARG;
  if (d->out_dtype == MTT_SPLIT && !d->out_lo) return MTT_E_BADARG;
  hipLaunchKernelGGL(modulate_kernel, dim3(grid_for((int64_t)d->B * d->h * d->w * (d->C / 8))), dim3(256), 0, S_, *d);
  return LAUNCH_OK();
}

extern "C" int mtt_ctr_mix(const mtt_ctr_desc* d, void* stream) {
  if (!d || !d->fea || !d->out || !d->wmix || d->T <= 0 || d->T > CTR_MAXT || (d->ld % 8)) return MTT_E_BADARG;
  if (d->out_dtype != MTT_F32 && (d->out_dtype != MTT_BF16 || d->accumulate)) return MTT_E_UNSUPPORTED;
  hipLaunchKernelGGL(ctr_mix_kernel, dim3(grid_for((int64_t)d->B * d->rows_per_b * ((d->C + 7) / 8))), dim3(256), 0, S_, *d);
  return LAUNCH_OK();
}

extern "C" int mtt_bilinear_fwd(const mtt_resize_desc* d, void* stream) {
  if (!d || !d->in || !d->out || d->B <= 0 || d->C <= 0) return MTT_E_BADARG;
  if (d->Hout > 65535 || d->B > 65535) return MTT_E_UNSUPPORTED;
  if (d->out_nchw) {
    const int sc = d->Hin > 1 && d->Win > 1 && d->Hout % d->Hin == 0 && d->Hout / d->Hin == d->Wout / d->Win && d->Wout % d->Win == 0 ? d->Hout / d->Hin : 0;
    const unsigned gx = (unsigned)(((int64_t)d->Win * ((d->C + 3) / 4) + 255) / 256);
    const bool v4 = (d->ld_in % 4) == 0 && d->ld_in >= (d->C + 3) / 4 * 4 && !((uintptr_t)d->in & 15);
    if (sc == 4 && v4 && !((uintptr_t)d->out & 15)) hipLaunchKernelGGL(bilinear_fwd_nchw_int_kernel<4>, dim3(gx, d->Hout, d->B), dim3(256), 0, S_, *d);
    else if (sc == 2 && v4 && !((uintptr_t)d->out & 7)) hipLaunchKernelGGL(bilinear_fwd_nchw_int_kernel<2>, dim3(gx, d->Hout, d->B), dim3(256), 0, S_, *d);
    else hipLaunchKernelGGL(bilinear_fwd_nchw_kernel, dim3((unsigned)((d->Wout + 255) / 256), d->Hout, d->B), dim3(256), 0, S_, *d);
  } else {
    if ((d->ld_in % 8) || (d->ld_out % 8)) return MTT_E_ALIGN;
    const int64_t cols = (int64_t)d->Wout * ((d->C + 7) / 8);
    hipLaunchKernelGGL(bilinear_fwd_nhwc_kernel, dim3((unsigned)((cols + 255) / 256), d->Hout, d->B), dim3(256), 0, S_, *d);
  }
  return LAUNCH_OK();
}
extern "C" int mtt_bilinear_bwd(const mtt_resize_desc* d, void* stream) {
  if (!d || !d->in || !d->out || d->B <= 0 || d->C <= 0) return MTT_E_BADARG;
  if (d->Hin > 65535 || d->B > 65535) return MTT_E_UNSUPPORTED;
  const int64_t cols = d->out_nchw ? (int64_t)d->Win * d->C : (int64_t)d->Win * ((d->C + 7) / 8);
  if (!d->out_nchw && (d->ld_in % 8)) return MTT_E_ALIGN;
  const int sc = d->out_nchw && d->Hin > 1 && d->Win > 1 && d->Hout % d->Hin == 0 && d->Wout % d->Win == 0 && d->Hout / d->Hin == d->Wout / d->Win &&
                 !((uintptr_t)d->in & 7) ? d->Hout / d->Hin : 0;
  // (an integer-scale NHWC twin with constant tent weights was built and measured in round 6: cfg4's 17 stage resizes 98.8 vs 100.4 ms over
  // five passes — they are bound by re-reading the fp32 gradient maps, not by the weight evaluation: profiles/r06_train_cfg4_b32_x3f_s.txt)
  const dim3 grid((unsigned)((cols + 255) / 256), d->Hin, d->B);
  if (sc == 4) hipLaunchKernelGGL(bilinear_bwd_nchw_int_kernel<4>, grid, dim3(256), 0, S_, *d);
  else if (sc == 2) hipLaunchKernelGGL(bilinear_bwd_nchw_int_kernel<2>, grid, dim3(256), 0, S_, *d);
  else hipLaunchKernelGGL(bilinear_bwd_kernel, grid, dim3(256), 0, S_, *d);
  return LAUNCH_OK();
}

static int colreduce_cfg(const mtt_bn_desc* d, int& nblk, int& rpb) {
  if (!d || d->rows <= 0 || d->C <= 0 || (d->ld % 8) || d->C > 2040) return MTT_E_BADARG;
  int64_t nb = (d->rows + 63) / 64; if (nb > 1024) nb = 1024; if (nb < 1) nb = 1;
  rpb = (int)((d->rows + nb - 1) / nb);
  nblk = (int)((d->rows + rpb - 1) / rpb);
  return 0;
}
static void bn_rowwise_cfg(const mtt_bn_desc* d, int& nblk, int& rpb) {
  const int C8 = (int)(d->ld / 8);
  const int lanes = C8 >= 256 ? 1 : 256 / C8;
  int64_t nb = (d->rows + 4 * lanes - 1) / (4 * lanes); if (nb > 4096) nb = 4096; if (nb < 1) nb = 1;   // >= 4 rows per lane
  rpb = (int)((d->rows + nb - 1) / nb);
  nblk = (int)((d->rows + rpb - 1) / rpb);
}
static int bn_batch(const mtt_bn_desc* d) { return d->Z > 1 ? d->Z : 1; }
extern "C" size_t mtt_bn_reduce_ws_floats(int64_t rows, int32_t C, int32_t Z) {
  mtt_bn_desc d = {}; d.rows = rows; d.C = C; d.ld = 8;
  int nblk, rpb; if (colreduce_cfg(&d, nblk, rpb)) return 0;
  return (size_t)(Z > 1 ? Z : 1) * nblk * 2 * ((C + 7) / 8 * 8);
}
extern "C" int mtt_bn_stats(const mtt_bn_desc* d, float* ws, void* stream) {
  int nblk, rpb; int e = colreduce_cfg(d, nblk, rpb); if (e) return e;
  if (!d->x || !d->mean_out || !d->m2_out || !ws) return MTT_E_BADARG;
  const int Z = bn_batch(d);
  hipLaunchKernelGGL(bn_reduce_kernel<0>, dim3(nblk, Z), dim3(256), 0, S_, *d, rpb, ws);
  hipLaunchKernelGGL(bn_final_kernel<0>, dim3((d->C + 31) / 32, Z), dim3(256), 0, S_, *d, rpb, nblk, ws);
  return LAUNCH_OK();
}
static void colsum_cfg(int64_t rows, int32_t cols, int& nblk, int& rpb, int& nchunk) {
  nchunk = (cols + 2047) / 2048;
  int64_t nb = (rows + 63) / 64;
  const int64_t cap = 768 / nchunk > 1 ? 768 / nchunk : 1;      // 768 row blocks saturate HBM; more only lengthen the second stage
  if (nb > cap) nb = cap;
  if (nb < 1) nb = 1;
  rpb = (int)((rows + nb - 1) / nb);
  nblk = (int)((rows + rpb - 1) / rpb);
}
extern "C" size_t mtt_colsum_ws_floats(int64_t rows, int32_t cols) {
  if (rows <= 0 || cols <= 0) return 0;
  int nblk, rpb, nchunk; colsum_cfg(rows, cols, nblk, rpb, nchunk);
  return (size_t)nblk * (size_t)((cols + 7) / 8 * 8);
}
extern "C" int mtt_colsum(const void* src, float* dst, int64_t rows, int32_t cols, int64_t ld, int src_dtype, float* ws, void* stream) {
  return mtt_colsum_batched(src, dst, rows, cols, ld, src_dtype, 1, 0, 0, ws, stream);
}
extern "C" int mtt_colsum_batched(const void* src, float* dst, int64_t rows, int32_t cols, int64_t ld, int src_dtype, int32_t Z, int64_t src_zs,
                                  int64_t dst_zs, float* ws, void* stream) {
  if (!src || !dst || !ws || rows <= 0 || cols <= 0 || Z <= 0 || Z > 65535 || (ld % 8) || ld < (cols + 7) / 8 * 8 || (src_zs % 8)) return MTT_E_BADARG;
  int nblk, rpb, nchunk; colsum_cfg(rows, cols, nblk, rpb, nchunk);
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(nblk, nchunk, Z), dim3(256), 0, S_, src, ws, rows, cols, ld, src_dtype, rpb, src_zs);
  hipLaunchKernelGGL(mtt_colsum_final_kernel, dim3((cols + 31) / 32, 1, Z), dim3(256), 0, S_, (const float*)ws, dst, cols, nblk, dst_zs);
  return LAUNCH_OK();
}
extern "C" int mtt_bn_bwd_reduce(const mtt_bn_desc* d, float* ws, void* stream) {
  int nblk, rpb; int e = colreduce_cfg(d, nblk, rpb); if (e) return e;
  if (!d->x || !d->dy || !d->dsum || !d->dsumxh || !d->mean || !d->rstd || !d->gamma || !d->beta || !ws) return MTT_E_BADARG;
  const int Z = bn_batch(d);
  hipLaunchKernelGGL(bn_reduce_kernel<2>, dim3(nblk, Z), dim3(256), 0, S_, *d, rpb, ws);
  hipLaunchKernelGGL(bn_final_kernel<2>, dim3((d->C + 31) / 32, Z), dim3(256), 0, S_, *d, rpb, nblk, ws);
  return LAUNCH_OK();
}
extern "C" int mtt_bn_apply(const mtt_bn_desc* d, void* stream) {
  if (!d || !d->x || !d->y || !d->mean || !d->rstd || !d->gamma || !d->beta || (d->ld % 8)) return MTT_E_BADARG;
  int nblk, rpb; bn_rowwise_cfg(d, nblk, rpb);
  hipLaunchKernelGGL(bn_rowwise_kernel<false>, dim3(nblk, bn_batch(d)), dim3(256), 0, S_, *d, rpb);
  return LAUNCH_OK();
}
extern "C" int mtt_bn_bwd_apply(const mtt_bn_desc* d, void* stream) {
  if (!d || !d->x || !d->dy || !d->dx || !d->mean || !d->rstd || !d->gamma || !d->beta || !d->dsum || !d->dsumxh || (d->ld % 8)) return MTT_E_BADARG;
  int nblk, rpb; bn_rowwise_cfg(d, nblk, rpb);
  hipLaunchKernelGGL(bn_rowwise_kernel<true>, dim3(nblk, bn_batch(d)), dim3(256), 0, S_, *d, rpb);
  return LAUNCH_OK();
}

extern "C" int mtt_split_cast(const float* src, void* hi, void* lo, int64_t rows, int64_t cols, int64_t lds_, int64_t ldd, void* stream) {
  if (!src || !hi || !lo || rows <= 0 || cols <= 0 || ldd < cols) return MTT_E_BADARG;
  if ((ldd % 8) || ((uintptr_t)hi & 15) || ((uintptr_t)lo & 15) || ((uintptr_t)src & 15)) return MTT_E_ALIGN;
  hipLaunchKernelGGL(split_cast_kernel, dim3(grid_for(rows * (ldd / 8))), dim3(256), 0, S_, src, (bf16_t*)hi, (bf16_t*)lo, rows, cols, lds_, ldd);
  return LAUNCH_OK();
}
extern "C" int mtt_segcopy_chunk(void) { return SEG_CHUNK; }
extern "C" int mtt_segcopy(const mtt_segcopy_desc* d, void* stream) {
  if (!d || !d->table || !d->chunk_seg || !d->chunk_off || d->n_chunks <= 0) return MTT_E_BADARG;
  hipLaunchKernelGGL(segcopy_kernel, dim3(d->n_chunks), dim3(256), 0, S_, *d);
  return LAUNCH_OK();
}

extern "C" int mtt_cast2d(const void* src, void* dst, int64_t rows, int64_t cols, int64_t lds_, int64_t ldd,
                          int src_dtype, int dst_dtype, int zero_pad_cols, void* stream) {
  if (!src || !dst || rows <= 0 || cols <= 0) return MTT_E_BADARG;
  hipLaunchKernelGGL(cast2d_kernel, dim3(grid_for(rows * (zero_pad_cols ? ldd : cols))), dim3(256), 0, S_, src, dst, rows, cols, lds_, ldd,
                     src_dtype, dst_dtype, zero_pad_cols);
  return LAUNCH_OK();
}
// Pixel shuffle of a ConvTranspose2d(k = 2, s = 2) computed as a plain GEMM: z [B*H*W, ldz] with column (dy*2+dx)*Co + co  ->
// out [B*2H*2W, ldo] NHWC (channels Co .. ldo-1 zeros).  One lane per 4 output channels of an output pixel; a quad's 4 source columns are
// consecutive inside one (dy, dx) block, read as scalars (the block offsets (dy*2+dx)*Co are not 16-byte aligned for odd Co).
__global__ __launch_bounds__(256) void pixshuf2_kernel(const void* z, void* out, int B, int H, int W, int Co, int64_t ldz, int64_t ldo, int zdt, int odt) {
  const int64_t quads = ldo >> 2;
  const int64_t total = (int64_t)B * 4 * H * W * quads;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int c0 = (int)(t % quads) * 4;
    const int64_t opix = t / quads;
    const int ox = (int)(opix % (2 * W));
    const int64_t r1 = opix / (2 * W);
    const int oy = (int)(r1 % (2 * H));
    const int64_t b = r1 / (2 * H);
    const int64_t zrow = (b * H + (oy >> 1)) * W + (ox >> 1);
    const int q = (oy & 1) * 2 + (ox & 1);
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = c0 + j < Co ? ld_elem(z, zrow * ldz + (int64_t)q * Co + c0 + j, zdt) : 0.f;
    if (odt == MTT_F32) *(float4*)((float*)out + opix * ldo + c0) = make_float4(v[0], v[1], v[2], v[3]);
    else *(u32x2*)((bf16_t*)out + opix * ldo + c0) = (u32x2){pack2(v[0], v[1]), pack2(v[2], v[3])};
  }
}
extern "C" int mtt_pixshuf2(const void* z, void* out, int32_t B, int32_t H, int32_t W, int32_t Co, int64_t ldz, int64_t ldo, int z_dtype, int out_dtype,
                            void* stream) {
  if (!z || !out || B <= 0 || H <= 0 || W <= 0 || Co <= 0 || ldz < 4 * (int64_t)Co || ldo < Co) return MTT_E_BADARG;
  if ((ldo % 8) || ((uintptr_t)out & 15)) return MTT_E_ALIGN;
  hipLaunchKernelGGL(pixshuf2_kernel, dim3(grid_for((int64_t)B * 4 * H * W * (ldo / 4))), dim3(256), 0, S_, z, out, B, H, W, Co, ldz, ldo, z_dtype, out_dtype);
  return LAUNCH_OK();
}
extern "C" int mtt_add_rows(const void* src, float* dst, int64_t rows, int32_t cols, int64_t lds_, int64_t ldd, int src_dtype,
                            float alpha, void* stream) {
  if (!src || !dst || rows <= 0 || cols <= 0) return MTT_E_BADARG;
  hipLaunchKernelGGL(add_rows_kernel, dim3(grid_for(rows * cols)), dim3(256), 0, S_, src, dst, rows, cols, lds_, ldd, src_dtype, alpha);
  return LAUNCH_OK();
}

extern "C" int mtt_modulate_bwd(const mtt_modulate_desc* d, const void* dout, float* dx, float* drawlog, float* drawchan, float* ws, void* stream) {
  if (!d || !d->x || !d->rawlog || !d->rawchan || !dout || !dx || !drawlog || !drawchan || !ws || (d->C % 32)) return MTT_E_BADARG;
  if (d->hg != 0 && d->hg != 64 && d->hg != 32) return MTT_E_UNSUPPORTED;      /* the head reduction of drawlog: 8 or 4 lanes */
  if (d->C % (d->hg > 0 ? d->hg : 64)) return MTT_E_BADARG;
  if (d->nh <= 0 || d->nw <= 0 || (d->h % d->nh) || (d->w % d->nw)) return MTT_E_BADARG;
  const int splits = modulate_bwd_splits(d);
  const int64_t n = (int64_t)d->B * d->T * d->nh * d->nw * d->C;
  dim3 grid((d->C / 8 + 31) / 32, splits, d->B * d->nh * d->nw);
  for (int tb = 0; tb < d->T; tb += CL_MAXT) {
    const int nt = d->T - tb < CL_MAXT ? d->T - tb : CL_MAXT;
#define MTT_MB(NT_) case NT_: hipLaunchKernelGGL(modulate_bwd_kernel<NT_>, grid, dim3(256), 0, S_, *d, dout, dx, drawlog, ws, n, tb); break;
    switch (nt) { MTT_MB(1) MTT_MB(2) MTT_MB(3) MTT_MB(4) MTT_MB(5) MTT_MB(6) MTT_MB(7) MTT_MB(8) }
#undef MTT_MB
  }
  hipLaunchKernelGGL(mtt_reduce_few_kernel, dim3(mtt_reduce_few_grid(n)), dim3(256), 0, S_, (const float*)ws, splits, n, drawchan, 0);
  return LAUNCH_OK();
}

extern "C" int mtt_chan_logits_bwd(const mtt_chanlogit_desc* d, const float* drawchan, void* dq, int dq_dtype, float* dxn, void* stream) {
  if (!d || !d->q || !d->xn || !drawchan || !dq || !dxn) return MTT_E_BADARG;
  if (d->nh <= 0 || d->nw <= 0 || (d->h % d->nh) || (d->w % d->nw) || (d->C % 8)) return MTT_E_BADARG;
  if (d->T > 2 * CL_MAXT) return MTT_E_UNSUPPORTED;
  if (d->dtype == MTT_SPLIT) return MTT_E_UNSUPPORTED;          // the backward reads ONE storage type (bf16 hi plane or fp32 rows)
  const int64_t toks = (int64_t)d->B * d->h * d->w;
  if (d->T <= CL_MAXT && ((d->w / d->nw) % 4) == 0 && (d->w % 4) == 0)
    hipLaunchKernelGGL(chanlogit_bwd_tok4_kernel, dim3((unsigned)((toks / 4 + 3) / 4)), dim3(256), 0, S_, *d, drawchan, dq, dq_dtype, dxn);
  else
    hipLaunchKernelGGL(chanlogit_bwd_kernel, dim3((unsigned)((toks + 3) / 4)), dim3(256), 0, S_, *d, drawchan, dq, dq_dtype, dxn);
  return LAUNCH_OK();
}

extern "C" int mtt_ctr_dw(const mtt_ctr_desc* d, const float* dout, float* dw, float* ws, void* stream) {
  if (!d || !d->fea || !dout || !dw || !ws || d->T <= 0 || d->T > CTR_MAXT || (d->ld % 8)) return MTT_E_BADARG;
  int rpb = 0;
  const int nb = ctr_dw_blocks(d, &rpb);
  const int64_t n = (int64_t)d->B * d->T * d->T;
  hipLaunchKernelGGL(ctr_dw_kernel, dim3(nb, d->B), dim3(256), 0, S_, *d, dout, ws, rpb);
  hipLaunchKernelGGL(mtt_reduce_few_kernel, dim3(mtt_reduce_few_grid(n)), dim3(256), 0, S_, (const float*)ws, nb, n, dw, 0);
  return LAUNCH_OK();
}

static void rscs_cfg(int64_t rows, int32_t cols, int& nblk, int& rpb) {
  const int panels = (cols + 255) / 256;
  int64_t nb = (rows + 63) / 64;
  const int64_t cap = 1024 / panels > 1 ? 1024 / panels : 1;
  if (nb > cap) nb = cap;
  if (nb < 1) nb = 1;
  rpb = (int)((rows + nb - 1) / nb);
  nblk = (int)((rows + rpb - 1) / rpb);
}
extern "C" size_t mtt_rowscale_cast_colsum_ws_floats(int64_t rows, int32_t cols) {
  if (rows <= 0 || cols <= 0) return 0;
  int nblk, rpb; rscs_cfg(rows, cols, nblk, rpb);
  return (size_t)nblk * (size_t)((cols + 7) / 8 * 8);
}
extern "C" int mtt_rowscale_cast_colsum(const void* src, void* dst, int64_t rows, int32_t cols, int64_t lds_, int64_t ldd, int src_dtype,
                                        int dst_dtype, const float* rowscale, int32_t mb, int32_t n_prompt, float* colsum_out, float* ws,
                                        void* stream) {
  if (!src || !dst || !colsum_out || !ws || rows <= 0 || cols <= 0 || rows >= (1ll << 31)) return MTT_E_BADARG;
  if ((cols % 8) || (lds_ % 8) || (ldd % 8) || (((uintptr_t)src | (uintptr_t)dst) & 15)) return MTT_E_ALIGN;
  int nblk, rpb; rscs_cfg(rows, cols, nblk, rpb);
  hipLaunchKernelGGL(rowscale_cast_colsum_kernel, dim3(nblk, (cols + 255) / 256), dim3(256), 0, S_, src, dst, (int)rows, cols / 8, lds_, ldd,
                     src_dtype, dst_dtype, rowscale, mb, n_prompt, ws, rpb);
  hipLaunchKernelGGL(mtt_colsum_final_kernel, dim3((cols + 31) / 32, 1, 1), dim3(256), 0, S_, (const float*)ws, colsum_out, cols, nblk, (int64_t)0);
  return LAUNCH_OK();
}
extern "C" int mtt_rowscale_cast(const void* src, void* dst, int64_t rows, int32_t cols, int64_t lds_, int64_t ldd, int src_dtype,
                                 int dst_dtype, const float* rowscale, int32_t mb, int32_t n_prompt, void* stream) {
  if (!src || !dst || rows <= 0 || cols <= 0) return MTT_E_BADARG;
  if ((cols % 8) == 0 && (lds_ % 8) == 0 && (ldd % 8) == 0 && rows * (cols / 8) < (1ll << 31) && !(((uintptr_t)src | (uintptr_t)dst) & 15)) {
    hipLaunchKernelGGL(rowscale_cast_vec_kernel, dim3(grid_for(rows * (cols / 8))), dim3(256), 0, S_, src, dst, (int)rows, cols / 8, lds_, ldd,
                       src_dtype, dst_dtype, rowscale, mb, n_prompt);
    return LAUNCH_OK();
  }
  hipLaunchKernelGGL(rowscale_cast_kernel, dim3(grid_for(rows * cols)), dim3(256), 0, S_, src, dst, rows, cols, lds_, ldd, src_dtype,
                     dst_dtype, rowscale, mb, n_prompt);
  return LAUNCH_OK();
}


// ---- cross-task reweighting weights: per-task 1x1-conv MLP over the head dimension of the prompt<->prompt raw logits (taskprompter.py:482-484)
namespace {
constexpr int CTRW_MAXH = 32;

// one thread per (b, t, s): z[h] = rawlog[b, h, t, s] -> wmix[b, t, s]; bwd (DZ): dz[h] -> drawlog[b, h, t, s]
template <bool DZ>
__global__ __launch_bounds__(256) void ctrw_item_kernel(const mtt_ctrw_desc d, const float* dwmix, float* drawlog) {
  const int item = blockIdx.x * 256 + threadIdx.x;
  const int T = d.T, nH = d.nH;
  if (item >= d.B * T * T) return;
  const int s = item % T, t = (item / T) % T, b = item / (T * T);
  float z[CTRW_MAXH];
#pragma unroll
  for (int h = 0; h < CTRW_MAXH; ++h)
    if (h < nH) z[h] = d.rawlog[(((int64_t)b * nH + h) * T + t) * d.N + s];
  const float* w0 = d.w0 + (int64_t)t * nH * nH;
  float out = DZ ? 0.f : d.b2[t];
  float dz[CTRW_MAXH];
  const float g = DZ ? dwmix[item] : 0.f;
  if (DZ) {
#pragma unroll
    for (int h = 0; h < CTRW_MAXH; ++h) dz[h] = 0.f;
  }
  for (int j = 0; j < nH; ++j) {
    float pre = d.b0[t * nH + j];
#pragma unroll
    for (int h = 0; h < CTRW_MAXH; ++h)
      if (h < nH) pre = fmaf(w0[j * nH + h], z[h], pre);
    if (!DZ) {
      out = fmaf(d.w2[t * nH + j], gelu_f(pre), out);
    } else {
      const float dpre = g * d.w2[t * nH + j] * gelu_grad_f(pre);
#pragma unroll
      for (int h = 0; h < CTRW_MAXH; ++h)
        if (h < nH) dz[h] = fmaf(dpre, w0[j * nH + h], dz[h]);
    }
  }
  if (!DZ) {
    d.wmix[item] = out;
  } else {
#pragma unroll
    for (int h = 0; h < CTRW_MAXH; ++h)
      if (h < nH) drawlog[(((int64_t)b * nH + h) * T + t) * d.N + s] = dz[h];
  }
}

// one workgroup per (t, j): the (b, s) items in thread order, per-thread partial sums, then a fixed-order tree through LDS
__global__ __launch_bounds__(256) void ctrw_wgrad_kernel(const mtt_ctrw_desc d, const float* dwmix, float* dw0, float* db0, float* dw2, float* db2) {
  __shared__ float red[256];
  const int T = d.T, nH = d.nH;
  const int j = blockIdx.x % nH, t = blockIdx.x / nH;
  const float* w0 = d.w0 + ((int64_t)t * nH + j) * nH;
  float aw0[CTRW_MAXH];
#pragma unroll
  for (int h = 0; h < CTRW_MAXH; ++h) aw0[h] = 0.f;
  float ab0 = 0.f, aw2 = 0.f, ab2 = 0.f;
  const int items = d.B * T;
  for (int it = threadIdx.x; it < items; it += 256) {
    const int s = it % T, b = it / T;
    float z[CTRW_MAXH];
    float pre = d.b0[t * nH + j];
#pragma unroll
    for (int h = 0; h < CTRW_MAXH; ++h)
      if (h < nH) {
        z[h] = d.rawlog[(((int64_t)b * nH + h) * T + t) * d.N + s];
        pre = fmaf(w0[h], z[h], pre);
      }
    const float g = dwmix[((int64_t)b * T + t) * T + s];
    const float dpre = g * d.w2[t * nH + j] * gelu_grad_f(pre);
    aw2 = fmaf(g, gelu_f(pre), aw2);
    ab2 += g;
    ab0 += dpre;
#pragma unroll
    for (int h = 0; h < CTRW_MAXH; ++h)
      if (h < nH) aw0[h] = fmaf(dpre, z[h], aw0[h]);
  }
  auto block_sum = [&](float v) {
    red[threadIdx.x] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
    const float r = red[0];
    __syncthreads();
    return r;
  };
#pragma unroll
  for (int h = 0; h < CTRW_MAXH; ++h)
    if (h < nH) {
      const float r = block_sum(aw0[h]);
      if (threadIdx.x == 0) dw0[((int64_t)t * nH + j) * nH + h] = r;
    }
  const float rb0 = block_sum(ab0), rw2 = block_sum(aw2), rb2 = block_sum(ab2);
  if (threadIdx.x == 0) {
    db0[t * nH + j] = rb0;
    dw2[t * nH + j] = rw2;
    if (j == 0) db2[t] = rb2;
  }
}

int ctrw_check(const mtt_ctrw_desc* d) {
  if (!d || !d->rawlog || !d->w0 || !d->b0 || !d->w2 || !d->b2 || d->B <= 0 || d->T <= 0 || d->nH <= 0 || d->N < d->T) return MTT_E_BADARG;
  if (d->nH > CTRW_MAXH || (int64_t)d->B * d->T * d->T >= (1ll << 31)) return MTT_E_UNSUPPORTED;
  return 0;
}
}  // namespace

extern "C" int mtt_ctr_weights(const mtt_ctrw_desc* d, void* stream) {
  if (int e = ctrw_check(d)) return e;
  if (!d->wmix) return MTT_E_BADARG;
  const int items = d->B * d->T * d->T;
  hipLaunchKernelGGL(ctrw_item_kernel<false>, dim3((items + 255) / 256), dim3(256), 0, S_, *d, (const float*)nullptr, (float*)nullptr);
  return LAUNCH_OK();
}
extern "C" int mtt_ctr_weights_bwd(const mtt_ctrw_desc* d, const float* dwmix, float* drawlog, float* dw0, float* db0, float* dw2, float* db2,
                                   void* stream) {
  if (int e = ctrw_check(d)) return e;
  if (!dwmix || !drawlog || !dw0 || !db0 || !dw2 || !db2) return MTT_E_BADARG;
  const int items = d->B * d->T * d->T;
  hipLaunchKernelGGL(ctrw_item_kernel<true>, dim3((items + 255) / 256), dim3(256), 0, S_, *d, dwmix, drawlog);
  hipLaunchKernelGGL(ctrw_wgrad_kernel, dim3(d->T * d->nH), dim3(256), 0, S_, *d, dwmix, dw0, db0, dw2, db2);
  return LAUNCH_OK();
}
