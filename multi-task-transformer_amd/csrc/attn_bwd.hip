// mtt_attn_bwd: flash-style backward of mtt_attn_fwd (bf16 storage, head_dim 64).  The N x N probabilities are
// recomputed tile by tile from q, k and the forward's log-sum-exp; nothing of size N x N reaches HBM and no atomics
// are used: two kernels own disjoint outputs.
//
//   dsum[b,h,i] = sum_d dO[i,d] * O[i,d]                                   (attn_dsum_kernel)
//   dQ kernel : block = 64 query rows of one (b, h); loops over key tiles:
//                 S = Q K^T, P = exp(scale*S - lse), dP = dO V^T, dS = scale * P * (dP - dsum) (+ drawlog on prompt rows),
//                 dQ += dS K
//   dKV kernel: block = 64 keys of one (b, h); loops over query tiles, everything transposed so that the key rows are
//                 the MFMA A operand held in registers:
//                 S^T = K Q^T, P^T, dP^T = V dO^T, dV += P^T dO, dK += dS^T Q
// Tiles are staged like the forward: 128 threads load 4-row x 8-column units and store them row-major AND transposed
// (in-register 4x8 transposes), the other 128 threads stage the second operand.  P^T / dS tiles go through a wave-private
// LDS tile to turn the MFMA C layout into an A fragment.  drawlog is the gradient of the forward's UNSCALED prompt-row
// logits side channel (rawlog), added to dS of the first T query rows.
#include "mtt_device.h"

namespace {

constexpr int KV = 64, HD = 64;
constexpr int KTILE = KV * HD * 2;   // 8 KiB per bf16 tile
constexpr int PT = 16 * 64 * 2;      // wave-private 16 x 64 bf16 tile
constexpr float LOG2E = 1.4426950408889634f;

struct BwdP {
  const bf16_t* qkv; const bf16_t* dout; const float* lse; const float* dsum; const float* drawlog; bf16_t* dqkv;
  int B, N, nH, T; float scale;
};

__global__ __launch_bounds__(256) void attn_dsum_kernel(const bf16_t* out, const bf16_t* dout, float* dsum, int B, int N, int nH) {
  const int C8 = nH * 8;
  const int64_t total = (int64_t)B * N * C8;
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float s = 0.f;
  if (t < total) {
    const u32x4 a = *(const u32x4*)(out + t * 8), g = *(const u32x4*)(dout + t * 8);
#pragma unroll
    for (int j = 0; j < 4; ++j) s += lo_of(a[j]) * lo_of(g[j]) + hi_of(a[j]) * hi_of(g[j]);
  }
  s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
  if (t < total && (threadIdx.x & 7) == 0) {
    const int64_t tok = t / C8;
    const int h = (int)(t % C8) >> 3;
    const int b = (int)(tok / N), n = (int)(tok % N);
    dsum[((int64_t)b * nH + h) * N + n] = s;
  }
}

MTT_DEV void store_tile_elem(unsigned char* Pw, int row, int col, float v) {
  *(bf16_t*)(Pw + lds_off(row, col >> 3) + (col & 7) * 2) = f2bf(v);
}

// --------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(const BwdP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int STAGE = 3 * KTILE;                  // K, K^T, V
  const int nqb = (p.N + 63) / 64;                            // XCD-aware 1-D grid: one head's blocks share an XCD's L2
  const int wi = xcd_remap(blockIdx.x, gridDim.x);
  const int qb = wi % nqb, bh_ = wi / nqb;
  const int h = bh_ % p.nH, b = bh_ / p.nH;
  const int N = p.N, C = p.nH * HD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int64_t tok0 = (int64_t)b * N;
  const int64_t bh = (int64_t)b * p.nH + h;
  unsigned char* const Pw = smem + 2 * STAGE + wave * PT;

  u32x4 qf[2], gf[2], dummy;
  {
    const int qrow = qb * 64 + wave * 16 + li;
    const bool ok = qrow < N;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      Raw8<false> r;
      load8_raw<false>(p.qkv, (tok0 + qrow) * 3 * C + h * HD + kh * 32 + lg * 8, ok, r);
      cvt8<false, false>(ok, r, qf[kh], dummy);
      load8_raw<false>(p.dout, (tok0 + qrow) * C + h * HD + kh * 32 + lg * 8, ok, r);
      cvt8<false, false>(ok, r, gf[kh], dummy);
    }
  }
  float lse2[4], Dr[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = qb * 64 + wave * 16 + lg * 4 + r;
    const bool ok = row < N;
    lse2[r] = ok ? p.lse[bh * N + row] * LOG2E : 0.f;
    Dr[r] = ok ? p.dsum[bh * N + row] : 0.f;
  }

  const bool isK = tid < 128;
  const int kq = tid & 15, rb = (tid >> 4) & 7;
  Raw8<false> raw[4];
  unsigned okm = 0;
  auto stage_load = [&](int kv0) {
    okm = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int key = kv0 + kq * 4 + i;
      const bool ok = key < N;
      okm |= (ok ? 1u : 0u) << i;
      load8_raw<false>(p.qkv, (tok0 + key) * 3 * C + (isK ? C : 2 * C) + h * HD + rb * 8, ok, raw[i]);
    }
  };
  auto stage_store = [&](unsigned char* st) {
    u32x4 sh[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) cvt8<false, false>((okm >> i) & 1u, raw[i], sh[i], dummy);
    unsigned char* rowmajor = isK ? st : st + 2 * KTILE;
#pragma unroll
    for (int i = 0; i < 4; ++i) *(u32x4*)(rowmajor + lds_off(kq * 4 + i, rb)) = sh[i];
    if (isK) {
      u32x2 piece[8];
      transpose4x8(sh, piece);
#pragma unroll
      for (int j = 0; j < 8; ++j) *(u32x2*)(st + KTILE + lds_off(rb * 8 + j, kq >> 1) + (kq & 1) * 8) = piece[j];
    }
  };

  f32x4 dq[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) dq[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float sc2 = p.scale * LOG2E;
  const bool add_raw = p.drawlog != nullptr && p.T > 0 && qb == 0 && wave == 0;

  const int nkv = (N + KV - 1) / KV;
  stage_load(0);
  stage_store(smem);
  __syncthreads();

  for (int j = 0; j < nkv; ++j) {
    const bool more = j + 1 < nkv;
    if (more) stage_load((j + 1) * KV);
    const unsigned char* st = smem + (j & 1) * STAGE;
    const unsigned char* Kh = st;
    const unsigned char* Kt = st + KTILE;
    const unsigned char* Vh = st + 2 * KTILE;
    const int kv0 = j * KV;

    f32x4 s[4], dp[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      s[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      dp[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
        const u32x4 kf = *(const u32x4*)(Kh + lds_off(nt * 16 + li, kh * 4 + lg));
        const u32x4 vf = *(const u32x4*)(Vh + lds_off(nt * 16 + li, kh * 4 + lg));
        s[nt] = mfma16(qf[kh], kf, s[nt]);
        dp[nt] = mfma16(gf[kh], vf, dp[nt]);
      }
    }
    const bool full_tile = kv0 + KV <= N;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int key = kv0 + nt * 16 + li;
      const bool kok = full_tile || key < N;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pv = kok ? __builtin_amdgcn_exp2f(s[nt][r] * sc2 - lse2[r]) : 0.f;
        float ds = pv * (dp[nt][r] - Dr[r]) * p.scale;
        if (add_raw) {
          const int row = lg * 4 + r;
          if (row < p.T && key < N) ds += p.drawlog[(bh * p.T + row) * N + key];
        }
        store_tile_elem(Pw, lg * 4 + r, nt * 16 + li, ds);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const u32x4 df = *(const u32x4*)(Pw + lds_off(li, ks * 4 + lg));
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const u32x4 kt = *(const u32x4*)(Kt + lds_off(dt * 16 + li, ks * 4 + lg));
        dq[dt] = mfma16(df, kt, dq[dt]);
      }
    }
    if (more) stage_store(smem + ((j + 1) & 1) * STAGE);
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int qrow = qb * 64 + wave * 16 + lg * 4 + r;
    if (qrow >= N) continue;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) p.dqkv[(tok0 + qrow) * 3 * C + h * HD + dt * 16 + li] = f2bf(dq[dt][r]);
  }
}

// --------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(const BwdP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int STAGE = 4 * KTILE;                  // Q, Q^T, dO, dO^T
  const int nkb = (p.N + 63) / 64;
  const int wi = xcd_remap(blockIdx.x, gridDim.x);
  const int kb = wi % nkb, bh_ = wi / nkb;
  const int h = bh_ % p.nH, b = bh_ / p.nH;
  const int N = p.N, C = p.nH * HD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int64_t tok0 = (int64_t)b * N;
  const int64_t bh = (int64_t)b * p.nH + h;
  unsigned char* const Pw = smem + 2 * STAGE + wave * PT;

  u32x4 kf[2], vf[2], dummy;
  {
    const int krow = kb * 64 + wave * 16 + li;
    const bool ok = krow < N;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      Raw8<false> r;
      load8_raw<false>(p.qkv, (tok0 + krow) * 3 * C + C + h * HD + kh * 32 + lg * 8, ok, r);
      cvt8<false, false>(ok, r, kf[kh], dummy);
      load8_raw<false>(p.qkv, (tok0 + krow) * 3 * C + 2 * C + h * HD + kh * 32 + lg * 8, ok, r);
      cvt8<false, false>(ok, r, vf[kh], dummy);
    }
  }

  const bool isQ = tid < 128;
  const int kq = tid & 15, rb = (tid >> 4) & 7;
  Raw8<false> raw[4];
  unsigned okm = 0;
  auto stage_load = [&](int q0) {
    okm = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = q0 + kq * 4 + i;
      const bool ok = row < N;
      okm |= (ok ? 1u : 0u) << i;
      if (isQ) load8_raw<false>(p.qkv, (tok0 + row) * 3 * C + h * HD + rb * 8, ok, raw[i]);
      else load8_raw<false>(p.dout, (tok0 + row) * C + h * HD + rb * 8, ok, raw[i]);
    }
  };
  auto stage_store = [&](unsigned char* st) {
    u32x4 sh[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) cvt8<false, false>((okm >> i) & 1u, raw[i], sh[i], dummy);
    unsigned char* base = isQ ? st : st + 2 * KTILE;
#pragma unroll
    for (int i = 0; i < 4; ++i) *(u32x4*)(base + lds_off(kq * 4 + i, rb)) = sh[i];
    u32x2 piece[8];
    transpose4x8(sh, piece);
#pragma unroll
    for (int j = 0; j < 8; ++j) *(u32x2*)(base + KTILE + lds_off(rb * 8 + j, kq >> 1) + (kq & 1) * 8) = piece[j];
  };

  f32x4 dk[4], dv[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) { dk[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  const float sc2 = p.scale * LOG2E;
  const int key0 = kb * 64 + wave * 16 + lg * 4;     // this lane's 4 key rows (C layout): key0 + r

  const int nq = (N + 63) / 64;
  stage_load(0);
  stage_store(smem);
  __syncthreads();

  for (int j = 0; j < nq; ++j) {
    const bool more = j + 1 < nq;
    if (more) stage_load((j + 1) * 64);
    const unsigned char* st = smem + (j & 1) * STAGE;
    const unsigned char* Qh = st;
    const unsigned char* Qt = st + KTILE;
    const unsigned char* Gh = st + 2 * KTILE;
    const unsigned char* Gt = st + 3 * KTILE;
    const int q0 = j * 64;

    float lse2c[4], Dc[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int qc = q0 + nt * 16 + li;
      const bool ok = qc < N;
      lse2c[nt] = ok ? p.lse[bh * N + qc] * LOG2E : 0.f;
      Dc[nt] = ok ? p.dsum[bh * N + qc] : 0.f;
    }
    f32x4 s[4], dp[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      s[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      dp[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
        const u32x4 qfr = *(const u32x4*)(Qh + lds_off(nt * 16 + li, kh * 4 + lg));
        const u32x4 gfr = *(const u32x4*)(Gh + lds_off(nt * 16 + li, kh * 4 + lg));
        s[nt] = mfma16(kf[kh], qfr, s[nt]);
        dp[nt] = mfma16(vf[kh], gfr, dp[nt]);
      }
    }
    // P^T -> wave tile, dV += P^T dO
    const bool full_tile = q0 + 64 <= N;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const bool qok = full_tile || (q0 + nt * 16 + li) < N;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pv = qok ? __builtin_amdgcn_exp2f(s[nt][r] * sc2 - lse2c[nt]) : 0.f;
        s[nt][r] = pv;
        store_tile_elem(Pw, lg * 4 + r, nt * 16 + li, pv);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const u32x4 pf = *(const u32x4*)(Pw + lds_off(li, ks * 4 + lg));
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const u32x4 gt = *(const u32x4*)(Gt + lds_off(dt * 16 + li, ks * 4 + lg));
        dv[dt] = mfma16(pf, gt, dv[dt]);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // dS^T -> wave tile, dK += dS^T Q
    const bool add_raw = p.drawlog != nullptr && j == 0 && p.T > 0;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int qc = q0 + nt * 16 + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float ds = s[nt][r] * (dp[nt][r] - Dc[nt]) * p.scale;
        if (add_raw && qc < p.T && key0 + r < N) ds += p.drawlog[(bh * p.T + qc) * N + key0 + r];
        store_tile_elem(Pw, lg * 4 + r, nt * 16 + li, ds);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const u32x4 df = *(const u32x4*)(Pw + lds_off(li, ks * 4 + lg));
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const u32x4 qt = *(const u32x4*)(Qt + lds_off(dt * 16 + li, ks * 4 + lg));
        dk[dt] = mfma16(df, qt, dk[dt]);
      }
    }
    if (more) stage_store(smem + ((j + 1) & 1) * STAGE);
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int krow = key0 + r;
    if (krow >= N) continue;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      p.dqkv[(tok0 + krow) * 3 * C + C + h * HD + dt * 16 + li] = f2bf(dk[dt][r]);
      p.dqkv[(tok0 + krow) * 3 * C + 2 * C + h * HD + dt * 16 + li] = f2bf(dv[dt][r]);
    }
  }
}

}  // namespace

extern "C" int mtt_attn_bwd(const mtt_attn_desc* d, const void* dout, const float* drawlog, void* dqkv, float* dsum, void* stream) {
  if (!d || !d->qkv || !d->out || !d->lse || !dout || !dqkv || !dsum) return MTT_E_BADARG;
  if (d->B <= 0 || d->N <= 0 || d->nH <= 0 || d->T < 0 || d->T > 16) return MTT_E_BADARG;
  if (d->dtype != MTT_BF16 || d->prec != MTT_PREC_BF16) return MTT_E_UNSUPPORTED;
  if (((uintptr_t)d->qkv | (uintptr_t)dout | (uintptr_t)d->out) & 15) return MTT_E_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  constexpr int smem_dq = 2 * 3 * KTILE + 4 * PT, smem_dkv = 2 * 4 * KTILE + 4 * PT;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)attn_bwd_dq_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, smem_dq);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, smem_dkv);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const int64_t chunks = (int64_t)d->B * d->N * d->nH * 8;
  hipLaunchKernelGGL(attn_dsum_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, s, (const bf16_t*)d->out, (const bf16_t*)dout,
                     dsum, d->B, d->N, d->nH);
  BwdP p{(const bf16_t*)d->qkv, (const bf16_t*)dout, d->lse, dsum, d->T > 0 ? drawlog : nullptr, (bf16_t*)dqkv, d->B, d->N, d->nH, d->T,
         d->scale};
  dim3 grid((unsigned)(((d->N + 63) / 64) * d->nH * d->B));
  hipLaunchKernelGGL(attn_bwd_dq_kernel, grid, dim3(256), smem_dq, s, p);
  hipLaunchKernelGGL(attn_bwd_dkv_kernel, grid, dim3(256), smem_dkv, s, p);
  return (int)hipGetLastError();
}
