// HBM-bound kernels specific to the InvPT decoder (InvPT/models/transformers/invpt.py, transformer_decoder.py):
// depthwise 3x3 stride-2 query projection, ceil-mode average pooling for keys/values, LayerNorm across all
// tasks' channels, the cross-stage attention-message fusion, and the gather half of ConvTranspose2d(k3,s2,p1,op1).
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

// y[z][b,oy,ox,c] = (sum_taps w[z][tap][c] * x[z][b, 2oy-1+ky, 2ox-1+kx, c]) * scale[z][c] + shift[z][c]
__global__ __launch_bounds__(256) void dwconv3s2_kernel(const mtt_dwconv_desc d) {
  const int C8 = d.ld >> 3;
  const int Ho = (d.H - 1) / 2 + 1, Wo = (d.W - 1) / 2 + 1;
  const int64_t total = (int64_t)d.Z * d.B * Ho * Wo * C8;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(t % C8);
    int64_t r = t / C8;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho); r /= Ho;
    const int b = (int)(r % d.B);
    const int z = (int)(r / d.B);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    const int64_t xin = ((int64_t)z * d.B + b) * d.H * d.W;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = 2 * oy - 1 + ky;
      if (iy < 0 || iy >= d.H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = 2 * ox - 1 + kx;
        if (ix < 0 || ix >= d.W) continue;
        float xv[8], wv[8];
        ld8(d.x, (xin + (int64_t)iy * d.W + ix) * d.ld + c8 * 8, d.dtype, xv);
        ld8(d.w, ((int64_t)z * 9 + ky * 3 + kx) * d.ld + c8 * 8, MTT_F32, wv);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += xv[j] * wv[j];
      }
    }
    if (d.scale) {
      float sc[8], sh[8];
      ld8(d.scale, (int64_t)z * d.ld + c8 * 8, MTT_F32, sc);
      ld8(d.shift, (int64_t)z * d.ld + c8 * 8, MTT_F32, sh);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = acc[j] * sc[j] + sh[j];
    }
    st8(d.y, ((((int64_t)z * d.B + b) * Ho + oy) * Wo + ox) * d.ld + c8 * 8, d.dtype, acc);
  }
}

// AvgPool2d(kernel = stride = k, padding 0, ceil_mode=True): divisor = number of in-bounds elements
__global__ __launch_bounds__(256) void avgpool_kernel(const mtt_pool_desc d) {
  const int C8 = d.ld >> 3;
  const int Ho = (d.H + d.k - 1) / d.k, Wo = (d.W + d.k - 1) / d.k;
  const int64_t total = (int64_t)d.B * Ho * Wo * C8;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(t % C8);
    int64_t r = t / C8;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho);
    const int b = (int)(r / Ho);
    const int y0 = oy * d.k, x0 = ox * d.k;
    const int y1 = y0 + d.k < d.H ? y0 + d.k : d.H, x1 = x0 + d.k < d.W ? x0 + d.k : d.W;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int y = y0; y < y1; ++y)
      for (int x = x0; x < x1; ++x) {
        float v[8];
        ld8(d.x, (((int64_t)b * d.H + y) * d.W + x) * d.ld + c8 * 8, d.dtype, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += v[j];
      }
    const float inv = 1.0f / (float)((y1 - y0) * (x1 - x0));
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] *= inv;
    st8(d.y, (((int64_t)b * Ho + oy) * Wo + ox) * d.ld + c8 * 8, d.dtype, acc);
  }
}

// LayerNorm over the concatenation of all T tasks' D channels of one pixel (invpt.py:482,526).
// x fp32 [T, rows, ld]; gamma/beta [T*D]; y [T, rows, ldy] (y_dtype).  One wave per pixel.
__global__ __launch_bounds__(256) void ln_mt_kernel(const mtt_lnmt_desc d) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= d.rows) return;
  float s = 0.f;
  for (int t = 0; t < d.T; ++t) {
    const float* x = d.x + ((int64_t)t * d.rows + row) * d.ldx;
    for (int c = lane; c < d.D; c += 64) s += x[c];
  }
  const float n = (float)(d.T * d.D);
  const float mean = wave_sum(s) / n;
  float q = 0.f;
  for (int t = 0; t < d.T; ++t) {
    const float* x = d.x + ((int64_t)t * d.rows + row) * d.ldx;
    for (int c = lane; c < d.D; c += 64) { const float a = x[c] - mean; q += a * a; }
  }
  const float rstd = rsqrtf(wave_sum(q) / n + d.eps);
  for (int t = 0; t < d.T; ++t) {
    const float* x = d.x + ((int64_t)t * d.rows + row) * d.ldx;
    const int64_t o = ((int64_t)t * d.rows + row) * d.ldy;
    for (int c = lane; c < d.ldy; c += 64)
      st_elem(d.y, o + c, d.y_dtype, c < d.D ? (x[c] - mean) * rstd * d.gamma[t * d.D + c] + d.beta[t * d.D + c] : 0.f);
  }
}

// Attention message passing (invpt.py:208-229): cur [B, heads, Q, K] (current stage scores, Q = T*qh*qw),
// prev [B, heads, Q/4, K] (previous stage, per task on a (qh/2 x qw/2) grid) is bilinearly upsampled x2 per task and
// fused:  out[b, ho, q, k] = bias[ho] + sum_h W[ho, h] * cur[b,h,q,k] + sum_h W[ho, heads+h] * up(prev)[b,h,q,k]
__global__ __launch_bounds__(256) void attn_msg_kernel(const mtt_attnmsg_desc d) {
  const int Qt = d.qh * d.qw, Q = d.T * Qt, sh = d.qh / 2, sw = d.qw / 2, Qp = d.T * sh * sw;
  const int64_t total = (int64_t)d.B * Q * d.K;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int k = (int)(i % d.K);
    int64_t r = i / d.K;
    const int q = (int)(r % Q);
    const int b = (int)(r / Q);
    const int t = q / Qt, p = q % Qt, y = p / d.qw, x = p % d.qw;
    // align_corners=False x2 upsample source coordinates
    float sy = (y + 0.5f) * 0.5f - 0.5f; if (sy < 0.f) sy = 0.f;
    float sx = (x + 0.5f) * 0.5f - 0.5f; if (sx < 0.f) sx = 0.f;
    int y0 = (int)sy, x0 = (int)sx;
    if (y0 > sh - 1) y0 = sh - 1;
    if (x0 > sw - 1) x0 = sw - 1;
    const int y1 = y0 < sh - 1 ? y0 + 1 : y0, x1 = x0 < sw - 1 ? x0 + 1 : x0;
    const float wy = sy - y0, wx = sx - x0;
    float vin[8];   // [cur heads..., prev heads...]   (heads <= 4)
    for (int h = 0; h < d.heads; ++h) {
      vin[h] = d.cur[(((int64_t)b * d.heads + h) * Q + q) * d.ldk + k];
      const float* pv = d.prev + (((int64_t)b * d.heads + h) * Qp + (int64_t)t * sh * sw) * d.ldkp;
      const float a00 = pv[(int64_t)(y0 * sw + x0) * d.ldkp + k], a01 = pv[(int64_t)(y0 * sw + x1) * d.ldkp + k];
      const float a10 = pv[(int64_t)(y1 * sw + x0) * d.ldkp + k], a11 = pv[(int64_t)(y1 * sw + x1) * d.ldkp + k];
      vin[d.heads + h] = (a00 * (1.f - wx) + a01 * wx) * (1.f - wy) + (a10 * (1.f - wx) + a11 * wx) * wy;
    }
    for (int ho = 0; ho < d.heads; ++ho) {
      float o = d.bias[ho];
      for (int h = 0; h < 2 * d.heads; ++h) o += d.w[ho * 2 * d.heads + h] * vin[h];
      d.out[(((int64_t)b * d.heads + ho) * Q + q) * d.ldk + k] = o;
    }
  }
}

// Backward of attn_msg_kernel w.r.t. everything but `prev` (whose gradient is the bilinear-backward gather of `dup`):
//   dcur[b,h,q,k] = sum_o W[o,h] dout[b,o,q,k]          dup[b,h,q,k] = sum_o W[o,heads+h] dout[b,o,q,k]
//   dw[o,h] += sum dout[b,o,..] cur[b,h,..]             dw[o,heads+h] += sum dout[b,o,..] up(prev)[b,h,..]        dbias[o] += sum dout[b,o,..]
// One pass over the score tensors; per-thread partial sums of the (<= 36) parameter gradients are reduced through LDS and added with
// one atomic per value per block (dw / dbias zeroed by the caller).
__global__ __launch_bounds__(256) void attn_msg_bwd_kernel(const mtt_attnmsg_desc d, const float* dout, float* dcur, float* dup, float* part) {
  __shared__ float red[4][40];
  const int Qt = d.qh * d.qw, Q = d.T * Qt, sh = d.qh / 2, sw = d.qw / 2, Qp = d.T * sh * sw;
  const int H = d.heads, H2 = 2 * d.heads;
  float pw[32], pb[4];                    // partial dw [H][2H] (H <= 4), dbias [H]
#pragma unroll
  for (int j = 0; j < 32; ++j) pw[j] = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) pb[j] = 0.f;
  const int64_t total = (int64_t)d.B * Q * d.K;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int k = (int)(i % d.K);
    int64_t r = i / d.K;
    const int q = (int)(r % Q);
    const int b = (int)(r / Q);
    const int t = q / Qt, p = q % Qt, y = p / d.qw, x = p % d.qw;
    float sy = (y + 0.5f) * 0.5f - 0.5f; if (sy < 0.f) sy = 0.f;
    float sx = (x + 0.5f) * 0.5f - 0.5f; if (sx < 0.f) sx = 0.f;
    int y0 = (int)sy, x0 = (int)sx;
    if (y0 > sh - 1) y0 = sh - 1;
    if (x0 > sw - 1) x0 = sw - 1;
    const int y1 = y0 < sh - 1 ? y0 + 1 : y0, x1 = x0 < sw - 1 ? x0 + 1 : x0;
    const float wy = sy - y0, wx = sx - x0;
    float vin[8], g[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      if (h >= H) break;
      const int64_t o = (((int64_t)b * H + h) * Q + q) * d.ldk + k;
      vin[h] = d.cur[o];
      g[h] = dout[o];
      const float* pv = d.prev + (((int64_t)b * H + h) * Qp + (int64_t)t * sh * sw) * d.ldkp;
      const float a00 = pv[(int64_t)(y0 * sw + x0) * d.ldkp + k], a01 = pv[(int64_t)(y0 * sw + x1) * d.ldkp + k];
      const float a10 = pv[(int64_t)(y1 * sw + x0) * d.ldkp + k], a11 = pv[(int64_t)(y1 * sw + x1) * d.ldkp + k];
      vin[H + h] = (a00 * (1.f - wx) + a01 * wx) * (1.f - wy) + (a10 * (1.f - wx) + a11 * wx) * wy;
    }
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      if (h >= H) break;
      float a = 0.f, u = 0.f;
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        if (o >= H) break;
        a += d.w[o * H2 + h] * g[o];
        u += d.w[o * H2 + H + h] * g[o];
      }
      const int64_t oidx = (((int64_t)b * H + h) * Q + q) * d.ldk + k;
      dcur[oidx] = a;
      dup[oidx] = u;
    }
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      if (o >= H) break;
      pb[o] += g[o];
#pragma unroll
      for (int h = 0; h < 8; ++h) {
        if (h >= H2) break;
        pw[o * 8 + h] += g[o] * vin[h];
      }
    }
  }
  // block reduction: wave shuffle, then the 4 waves through LDS
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < 32; ++j) { const float v = wave_sum(pw[j]); if (lane == 0) red[wave][j] = v; }
#pragma unroll
  for (int j = 0; j < 4; ++j) { const float v = wave_sum(pb[j]); if (lane == 0) red[wave][32 + j] = v; }
  __syncthreads();
  if (threadIdx.x < 36) {
    const float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    part[(int64_t)blockIdx.x * 36 + threadIdx.x] = v;      // 32 weight + 4 bias partials per block; mtt_reduce_many_kernel sums the blocks in order
  }
}

// dw [H, 2H] / dbias [H] from the 36 reduced sums (slot o * 8 + h / 32 + o)
__global__ void attn_msg_bwd_final_kernel(const float* sums, float* dw, float* dbias, int H) {
  const int H2 = 2 * H, i = threadIdx.x;
  if (i < 32) { const int o = i >> 3, h = i & 7; if (o < H && h < H2) dw[o * H2 + h] = sums[i]; }
  else if (i < 36 && i - 32 < H) dbias[i - 32] = sums[i];
}

// ConvTranspose2d(k=3, s=2, p=1, output_padding=1) gather: yall [B*H*W, 9*Cop] = x @ Wall^T (tap-major columns) ->
// out[b, oy, ox, co] = bias[co] + sum over taps with oy = 2*iy - 1 + ky, ox = 2*ix - 1 + kx
__global__ __launch_bounds__(256) void convt3s2_gather_kernel(const mtt_convt_desc d) {
  const int C8 = d.Cop >> 3, Ho = 2 * d.H, Wo = 2 * d.W;
  const int64_t total = (int64_t)d.B * Ho * Wo * C8;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(t % C8);
    int64_t r = t / C8;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho);
    const int b = (int)(r / Ho);
    float acc[8];
    ld8(d.bias, c8 * 8, MTT_F32, acc);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int ty = oy + 1 - ky;
      if (ty < 0 || (ty & 1) || (ty >> 1) >= d.H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int tx = ox + 1 - kx;
        if (tx < 0 || (tx & 1) || (tx >> 1) >= d.W) continue;
        float v[8];
        ld8(d.yall, (((int64_t)b * d.H + (ty >> 1)) * d.W + (tx >> 1)) * (9 * d.Cop) + (ky * 3 + kx) * d.Cop + c8 * 8, d.dtype, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += v[j];
      }
    }
    st8(d.out, (((int64_t)b * Ho + oy) * Wo + ox) * d.Cop + c8 * 8, d.out_dtype, acc);
  }
}


// ---- backward kernels ------------------------------------------------------------------------------------
// depthwise 3x3 s2: dx[z,b,iy,ix,c] = sum over taps with iy = 2*oy - 1 + ky of w[z,tap,c] * dy[z,b,oy,ox,c]
__global__ __launch_bounds__(256) void dwconv3s2_bwd_dx_kernel(const mtt_dwconv_desc d, const void* dy, void* dx) {
  const int C8 = d.ld >> 3;
  const int Ho = (d.H - 1) / 2 + 1, Wo = (d.W - 1) / 2 + 1;
  const int64_t total = (int64_t)d.Z * d.B * d.H * d.W * C8;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(t % C8);
    int64_t r = t / C8;
    const int ix = (int)(r % d.W); r /= d.W;
    const int iy = (int)(r % d.H); r /= d.H;
    const int b = (int)(r % d.B);
    const int z = (int)(r / d.B);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int ty = iy + 1 - ky;
      if (ty < 0 || (ty & 1) || (ty >> 1) >= Ho) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int tx = ix + 1 - kx;
        if (tx < 0 || (tx & 1) || (tx >> 1) >= Wo) continue;
        float g[8], wv[8];
        ld8(dy, ((((int64_t)z * d.B + b) * Ho + (ty >> 1)) * Wo + (tx >> 1)) * d.ld + c8 * 8, d.dtype, g);
        ld8(d.w, ((int64_t)z * 9 + ky * 3 + kx) * d.ld + c8 * 8, MTT_F32, wv);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += g[j] * wv[j];
      }
    }
    st8(dx, ((((int64_t)z * d.B + b) * d.H + iy) * d.W + ix) * d.ld + c8 * 8, d.dtype, acc);
  }
}
// dw[z,tap,c] = sum_{b,oy,ox} x[z,b,2oy-1+ky,2ox-1+kx,c] * dy[z,b,oy,ox,c].
// One workgroup per (z, tap, group of 64 channels): 8 chunk lanes (8 channels = 16 B each: a pixel's 128 contiguous bytes) x 32 pixel lanes
// that stride over the B * Ho * Wo output pixels; fixed-order reduction (xor shuffles over the 8 pixel lanes of a wave, then the 4 waves
// through LDS): deterministic, no atomics, no workspace.  (Round 1 looped over ALL pixels in one thread per (z, tap, 8 channels): 12 ms per
// call at the InvPT ViT-L sizes, 7.7 % of the cfg4 training step, profiles/r02_train_cfg4_b32_z.txt.)
__global__ __launch_bounds__(256) void dwconv3s2_bwd_dw_kernel(const mtt_dwconv_desc d, const void* dy, float* dw) {
  const int C8 = (int)(d.ld >> 3);
  const int Ho = (d.H - 1) / 2 + 1, Wo = (d.W - 1) / 2 + 1;
  const int groups = (C8 + 7) >> 3;
  const int grp = blockIdx.x % groups, tap = (blockIdx.x / groups) % 9, z = blockIdx.x / (groups * 9);
  const int ky = tap / 3, kx = tap % 3;
  const int cl = threadIdx.x & 7, pl = threadIdx.x >> 3;            // chunk lane, pixel lane (0..31)
  const int c8 = grp * 8 + cl;
  const bool cok = c8 < C8;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  const int npix = d.B * Ho * Wo;
  if (cok) {
    for (int p = pl; p < npix; p += 32) {
      const int ox = p % Wo, t = p / Wo;
      const int oy = t % Ho, b = t / Ho;
      const int iy = 2 * oy - 1 + ky, ix = 2 * ox - 1 + kx;
      if (iy < 0 || iy >= d.H || ix < 0 || ix >= d.W) continue;
      float xv[8], g[8];
      ld8(d.x, ((((int64_t)z * d.B + b) * d.H + iy) * d.W + ix) * d.ld + c8 * 8, d.dtype, xv);
      ld8(dy, ((int64_t)z * npix + p) * d.ld + c8 * 8, d.dtype, g);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = fmaf(xv[j], g[j], acc[j]);
    }
  }
  __shared__ float red[4][8][8];
  const int wave = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float v = acc[j];
    v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
    if ((threadIdx.x & 63) < 8) red[wave][cl][j] = v;
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const int c = threadIdx.x >> 3, j = threadIdx.x & 7;
    if (grp * 8 + c < C8)
      dw[((int64_t)z * 9 + tap) * d.ld + (grp * 8 + c) * 8 + j] = (red[0][c][j] + red[1][c][j]) + (red[2][c][j] + red[3][c][j]);
  }
}
// average-pool backward: dx[b,y,x,c] = dy[b, y/k, x/k, c] / (#in-bounds elements of that window)
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const mtt_pool_desc d, const void* dy, void* dx) {
  const int C8 = d.ld >> 3;
  const int Ho = (d.H + d.k - 1) / d.k, Wo = (d.W + d.k - 1) / d.k;
  const int64_t total = (int64_t)d.B * d.H * d.W * C8;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(t % C8);
    int64_t r = t / C8;
    const int x = (int)(r % d.W); r /= d.W;
    const int y = (int)(r % d.H);
    const int b = (int)(r / d.H);
    const int oy = y / d.k, ox = x / d.k;
    const int hh = (oy + 1) * d.k < d.H ? d.k : d.H - oy * d.k, ww = (ox + 1) * d.k < d.W ? d.k : d.W - ox * d.k;
    float g[8];
    ld8(dy, (((int64_t)b * Ho + oy) * Wo + ox) * d.ld + c8 * 8, d.dtype, g);
    const float inv = 1.0f / (float)(hh * ww);
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] *= inv;
    st8(dx, (((int64_t)b * d.H + y) * d.W + x) * d.ld + c8 * 8, d.dtype, g);
  }
}
// ConvTranspose gather backward: dyall[b,iy,ix,tap,co] = dout[b, 2iy-1+ky, 2ix-1+kx, co] (0 outside)
__global__ __launch_bounds__(256) void convt3s2_gather_bwd_kernel(const mtt_convt_desc d, const void* dout, void* dyall) {
  const int C8 = d.Cop >> 3, Ho = 2 * d.H, Wo = 2 * d.W;
  const int64_t total = (int64_t)d.B * d.H * d.W * 9 * C8;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(t % C8);
    int64_t r = t / C8;
    const int tap = (int)(r % 9); r /= 9;
    const int ix = (int)(r % d.W); r /= d.W;
    const int iy = (int)(r % d.H);
    const int b = (int)(r / d.H);
    const int oy = 2 * iy - 1 + tap / 3, ox = 2 * ix - 1 + tap % 3;
    float g[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] = 0.f;
    if (oy >= 0 && oy < Ho && ox >= 0 && ox < Wo) ld8(dout, (((int64_t)b * Ho + oy) * Wo + ox) * d.Cop + c8 * 8, d.out_dtype, g);
    st8(dyall, (((int64_t)b * d.H + iy) * d.W + ix) * (9 * d.Cop) + tap * d.Cop + c8 * 8, d.dtype, g);
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

extern "C" int mtt_dwconv3x3s2(const mtt_dwconv_desc* d, void* stream) {
  if (!d || !d->x || !d->w || !d->y || d->Z <= 0 || d->B <= 0 || (d->ld % 8) || (d->scale && !d->shift)) return MTT_E_BADARG;
  const int Ho = (d->H - 1) / 2 + 1, Wo = (d->W - 1) / 2 + 1;
  hipLaunchKernelGGL(dwconv3s2_kernel, dim3(grid_for((int64_t)d->Z * d->B * Ho * Wo * (d->ld / 8))), dim3(256), 0, S_, *d);
  return (int)hipGetLastError();
}
extern "C" int mtt_avgpool_ceil(const mtt_pool_desc* d, void* stream) {
  if (!d || !d->x || !d->y || d->B <= 0 || d->k <= 0 || (d->ld % 8)) return MTT_E_BADARG;
  const int Ho = (d->H + d->k - 1) / d->k, Wo = (d->W + d->k - 1) / d->k;
  hipLaunchKernelGGL(avgpool_kernel, dim3(grid_for((int64_t)d->B * Ho * Wo * (d->ld / 8))), dim3(256), 0, S_, *d);
  return (int)hipGetLastError();
}
extern "C" int mtt_layernorm_mt(const mtt_lnmt_desc* d, void* stream) {
  if (!d || !d->x || !d->y || !d->gamma || !d->beta || d->rows <= 0 || d->T <= 0 || d->D <= 0) return MTT_E_BADARG;
  hipLaunchKernelGGL(ln_mt_kernel, dim3((unsigned)((d->rows + 3) / 4)), dim3(256), 0, S_, *d);
  return (int)hipGetLastError();
}
extern "C" int mtt_attn_msg(const mtt_attnmsg_desc* d, void* stream) {
  if (!d || !d->cur || !d->prev || !d->out || !d->w || !d->bias || d->heads <= 0 || d->heads > 4 || (d->qh % 2) || (d->qw % 2)) return MTT_E_BADARG;
  hipLaunchKernelGGL(attn_msg_kernel, dim3(grid_for((int64_t)d->B * d->T * d->qh * d->qw * d->K)), dim3(256), 0, S_, *d);
  return (int)hipGetLastError();
}
static int64_t attn_msg_bwd_blocks(const mtt_attnmsg_desc* d) {
  int64_t g = ((int64_t)d->B * d->T * d->qh * d->qw * d->K + 255) / 256;
  return g > 2048 ? 2048 : (g < 1 ? 1 : g);
}
extern "C" size_t mtt_attn_msg_bwd_ws_floats(const mtt_attnmsg_desc* d) { return d ? (size_t)(attn_msg_bwd_blocks(d) + 1) * 36 : 0; }
extern "C" int mtt_attn_msg_bwd(const mtt_attnmsg_desc* d, const float* dout, float* dcur, float* dup, float* dw, float* dbias, float* ws, void* stream) {
  if (!d || !d->cur || !d->prev || !d->w || !dout || !dcur || !dup || !dw || !dbias || !ws || d->heads <= 0 || d->heads > 4 || (d->qh % 2) || (d->qw % 2))
    return MTT_E_BADARG;
  const int64_t g = attn_msg_bwd_blocks(d);
  float* sums = ws + g * 36;
  hipLaunchKernelGGL(attn_msg_bwd_kernel, dim3((unsigned)g), dim3(256), 0, S_, *d, dout, dcur, dup, ws);
  hipLaunchKernelGGL(mtt_reduce_many_kernel, dim3(36), dim3(256), 0, S_, (const float*)ws, (int)g, 36, sums, 1.0f, 0);
  hipLaunchKernelGGL(attn_msg_bwd_final_kernel, dim3(1), dim3(64), 0, S_, (const float*)sums, dw, dbias, d->heads);
  return (int)hipGetLastError();
}
extern "C" int mtt_convt3x3s2_gather(const mtt_convt_desc* d, void* stream) {
  if (!d || !d->yall || !d->out || !d->bias || d->B <= 0 || (d->Cop % 8)) return MTT_E_BADARG;
  hipLaunchKernelGGL(convt3s2_gather_kernel, dim3(grid_for((int64_t)d->B * 4 * d->H * d->W * (d->Cop / 8))), dim3(256), 0, S_, *d);
  return (int)hipGetLastError();
}

extern "C" int mtt_dwconv3x3s2_bwd(const mtt_dwconv_desc* d, const void* dy, void* dx, float* dw, void* stream) {
  if (!d || !d->x || !d->w || !dy || d->Z <= 0 || d->B <= 0 || (d->ld % 8)) return MTT_E_BADARG;
  if (dx) hipLaunchKernelGGL(dwconv3s2_bwd_dx_kernel, dim3(grid_for((int64_t)d->Z * d->B * d->H * d->W * (d->ld / 8))), dim3(256), 0, S_, *d, dy, dx);
  if (dw) hipLaunchKernelGGL(dwconv3s2_bwd_dw_kernel, dim3((unsigned)((int64_t)d->Z * 9 * ((d->ld / 8 + 7) / 8))), dim3(256), 0, S_, *d, dy, dw);
  return (int)hipGetLastError();
}
extern "C" int mtt_avgpool_ceil_bwd(const mtt_pool_desc* d, const void* dy, void* dx, void* stream) {
  if (!d || !dy || !dx || d->B <= 0 || d->k <= 0 || (d->ld % 8)) return MTT_E_BADARG;
  hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(grid_for((int64_t)d->B * d->H * d->W * (d->ld / 8))), dim3(256), 0, S_, *d, dy, dx);
  return (int)hipGetLastError();
}
extern "C" int mtt_convt3x3s2_gather_bwd(const mtt_convt_desc* d, const void* dout, void* dyall, void* stream) {
  if (!d || !dout || !dyall || d->B <= 0 || (d->Cop % 8)) return MTT_E_BADARG;
  hipLaunchKernelGGL(convt3s2_gather_bwd_kernel, dim3(grid_for((int64_t)d->B * d->H * d->W * 9 * (d->Cop / 8))), dim3(256), 0, S_, *d, dout, dyall);
  return (int)hipGetLastError();
}
