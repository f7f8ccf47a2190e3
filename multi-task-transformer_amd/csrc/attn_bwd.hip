// mtt_attn_bwd: flash-style backward of mtt_attn_fwd (bf16 storage, head_dim 64).  The N x N probabilities are recomputed
// tile by tile from q, k and the forward's log-sum-exp; nothing of size N x N reaches HBM and no atomics are used: two
// kernels own disjoint outputs.  Both use the "swapped product" form of attn_fast.hip: a score tile comes out of the MFMA
// in the C layout that IS the B fragment of the next product (reduction index permuted consistently on the A side, whose
// fragments are two 8-byte reads of a transposed LDS tile), so P and dS never go through LDS.
//
//   stat[b,h,0,i] = sum_d dO[i,d] * O[i,d],  stat[b,h,1,i] = lse[i] * log2(e)                (attn_stat_kernel)
//   dQ kernel : workgroup = 128 query rows (32 per wave, B fragments of Q and dO in registers); per 64-key tile
//                 S^T = K Q^T, dP^T = V dO^T, dS^T = P^T o (dP^T - D),   dQ^T += K^T dS^T
//   dKV kernel: workgroup = 128 keys (32 per wave, B fragments of K and V in registers); per 64-query tile
//                 S = Q K^T, dP = dO V^T,   dV^T += dO^T P,   dK^T += Q^T dS
// The softmax scale is applied once to the dQ / dK accumulators; drawlog (gradient of the forward's UNSCALED prompt-row
// logits) is therefore added to dS divided by the scale.
#include "mtt_device.h"
#include <type_traits>

namespace {

constexpr int HD = 64;
constexpr int KTILE = 64 * HD * 2;   // 8 KiB per bf16 tile
constexpr float LOG2E = 1.4426950408889634f;

struct BwdP {
  const bf16_t* qkv; const bf16_t* dout; const float* stat; const float* drawlog; bf16_t* dqkv;
  int B, N, nH, T, Np; float scale;
};

__global__ __launch_bounds__(256) void attn_stat_kernel(const bf16_t* out, const bf16_t* dout, const float* lse, float* stat, int B, int N,
                                                        int nH, int Np) {
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
    float* row = stat + ((int64_t)b * nH + h) * 2 * Np;
    row[n] = s;
    row[Np + n] = lse[((int64_t)b * nH + h) * N + n] * LOG2E;
    if (n == N - 1)
      for (int k = N; k < Np; ++k) { row[k] = 0.f; row[Np + k] = 0.f; }
  }
}

// A fragment whose reduction index follows the C-layout permutation of a 32-wide k step (see attn_fast.hip)
MTT_DEV u32x4 perm_frag(const unsigned char* tile, int row, int ks, int lg) {
  const u32x2 lo = *(const u32x2*)(tile + lds_off(row, 4 * ks + (lg >> 1)) + (lg & 1) * 8);
  const u32x2 hi = *(const u32x2*)(tile + lds_off(row, 4 * ks + 2 + (lg >> 1)) + (lg & 1) * 8);
  return (u32x4){lo[0], lo[1], hi[0], hi[1]};
}

// 4-row x 8-column units of a 64-row tile: 128 threads stage one operand row-major (and transposed when TR)
template <bool TR>
MTT_DEV void store_units(unsigned char* rowmajor, unsigned char* transposed, const u32x4 (&sh)[4], int kq, int rb) {
#pragma unroll
  for (int i = 0; i < 4; ++i) *(u32x4*)(rowmajor + lds_off(kq * 4 + i, rb)) = sh[i];
  if (TR) {
    u32x2 piece[8];
    transpose4x8(sh, piece);
#pragma unroll
    for (int j = 0; j < 8; ++j) *(u32x2*)(transposed + lds_off(rb * 8 + j, kq >> 1) + (kq & 1) * 8) = piece[j];
  }
}

// ---- VER 2 staging (see attn_fast.hip): tiles go HBM -> LDS by LDS-DMA as they sit in memory ([64 rows][8 chunks of 16 B]); the transposed
// A fragments (K^T, Q^T, dO^T) come from ds_read_b64_tr_b16 on the SAME row-major image, so no transposed copy is staged at all (the dQ
// kernel stages 16 KiB per tile instead of 24, the dK/dV kernel 16 instead of 32).  One swizzle serves both read kinds: 16-byte chunk c of
// row r sits at position c ^ x(r), x(r) = ((r >> 1) & 3) << 1 | ((r >> 3) & 1): the 16 rows of a ds_read_b128 fragment read hit 16
// different (row parity, position) pairs, and the 8 rows a 32-lane half of a transpose read touches hit 8 different 32-byte units.
__device__ __attribute__((aligned(32))) const unsigned g_bwd_zero_page[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
MTT_DEV int bwd_swz(int row) { return (((row >> 1) & 3) << 1) | ((row >> 3) & 1); }
MTT_DEV void bwd_glds16(const bf16_t* src, unsigned char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
template <int OFF>
MTT_DEV u32x2 bwd_ds_read_tr16(unsigned addr) {
  u32x2 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
  return r;
}
#define BWD_TRW(x) "+v"(x)
// waits for EVERY outstanding LDS read; the transpose-read destinations are in/out operands so that no use (and no register copy) of them
// can be placed above the wait — the asm loads are invisible to the compiler's own wait counting
#define BWD_WAIT_TR8(a, b)                                                                                                              \
  asm volatile("s_waitcnt lgkmcnt(0)" : BWD_TRW(a[0]), BWD_TRW(a[1]), BWD_TRW(a[2]), BWD_TRW(a[3]), BWD_TRW(b[0]), BWD_TRW(b[1]), BWD_TRW(b[2]), BWD_TRW(b[3]) :: "memory")
#define BWD_WAIT_TR16(a, b, c, d)                                                                                                       \
  asm volatile("s_waitcnt lgkmcnt(0)" : BWD_TRW(a[0]), BWD_TRW(a[1]), BWD_TRW(a[2]), BWD_TRW(a[3]), BWD_TRW(b[0]), BWD_TRW(b[1]), BWD_TRW(b[2]), BWD_TRW(b[3]), \
               BWD_TRW(c[0]), BWD_TRW(c[1]), BWD_TRW(c[2]), BWD_TRW(c[3]), BWD_TRW(d[0]), BWD_TRW(d[1]), BWD_TRW(d[2]), BWD_TRW(d[3]) :: "memory")

// --------------------------------------------------------------------------------------------------------
// VER 2 (default): LDS-DMA staging + transpose reads (above).  VER 1 (MTT_ATTN_FAST_V1): register-staged tiles with a transposed copy;
// VER 0 = the first form, MTT_ATTN_FAST_V0.
// VER 1 over VER 0: loop unrolled by the two LDS stages (immediate stage offsets), full tiles
// staged without per-row predicates, MFMA clusters at raised wave priority — the three changes that took 7 % off the forward kernel.
template <int VER>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(const BwdP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int STAGE = (VER == 2 ? 2 : 3) * KTILE;  // K, K^T, V (VER 2: K, V)
  const int nqb = (p.N + 127) / 128;                // XCD-aware 1-D grid: one head's blocks share an XCD's L2
  const int wi = xcd_remap(blockIdx.x, gridDim.x);
  const int qb = wi % nqb, bh_ = wi / nqb;
  const int h = bh_ % p.nH, b = bh_ / p.nH;
  const int N = p.N, C = p.nH * HD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int64_t tok0 = (int64_t)b * N;
  const int64_t bh = (int64_t)b * p.nH + h;
  const int q0 = qb * 128 + wave * 32;
  const bool active = q0 < N;

  u32x4 qf[2][2], gf[2][2], dummy;
  float lse2[2], Dq[2];
#pragma unroll
  for (int sub = 0; sub < 2; ++sub) {
    const int qrow = q0 + sub * 16 + li;
    const bool ok = qrow < N;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      Raw8<false> r;
      load8_raw<false>(p.qkv, (tok0 + qrow) * 3 * C + h * HD + kh * 32 + lg * 8, ok, r);
      cvt8<false, false>(ok, r, qf[sub][kh], dummy);
      load8_raw<false>(p.dout, (tok0 + qrow) * C + h * HD + kh * 32 + lg * 8, ok, r);
      cvt8<false, false>(ok, r, gf[sub][kh], dummy);
    }
    Dq[sub] = ok ? p.stat[bh * 2 * p.Np + qrow] : 0.f;
    lse2[sub] = ok ? p.stat[bh * 2 * p.Np + p.Np + qrow] : 0.f;
  }

  const bool isK = tid < 128;
  const int kq = tid & 15, rb = (tid >> 4) & 7;
  Raw8<false> raw[4];
  unsigned okm = 0;
  auto stage_load = [&](int kv0) {
    if (VER >= 1 && kv0 + 64 <= N) {                  // block-uniform: a full tile needs no per-key selects
      okm = 0xfu;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        raw[i].r0 = *(const u32x4*)(p.qkv + ((tok0 + kv0 + kq * 4 + i) * 3 * C + (isK ? C : 2 * C) + h * HD + rb * 8));
      return;
    }
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
    if (VER >= 1 && okm == 0xfu) {
#pragma unroll
      for (int i = 0; i < 4; ++i) sh[i] = raw[i].r0;
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) cvt8<false, false>((okm >> i) & 1u, raw[i], sh[i], dummy);
    }
    if (isK) store_units<true>(st, st + KTILE, sh, kq, rb);
    else store_units<false>(st + 2 * KTILE, nullptr, sh, kq, rb);
  };

  // VER 2: wave w moves key rows [16 w, 16 w + 16) of the K and of the V tile as two 1-KiB pieces each
  int dma_off[4], dma_row[2];
  uint64_t zpage = 0;
  unsigned taddr[4];                                 // K^T transpose-read addresses per d tile (stage, 32-key half, 4-key half: immediates)
  const unsigned char* faddr[2];                     // row fragment (ds_read_b128) addresses per kh (key tile, stage, K / V: immediates)
  if (VER == 2) {
    zpage = (uint64_t)(uintptr_t)g_bwd_zero_page;
    asm volatile("" : "+s"(zpage));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = wave * 16 + i * 8 + (lane >> 3), pc = lane & 7;
      dma_row[i] = r;
      dma_off[i] = r * 3 * C + C + ((pc ^ bwd_swz(r)) & 7) * 8;
      dma_off[2 + i] = r * 3 * C + 2 * C + ((pc ^ bwd_swz(r)) & 7) * 8;
    }
    const int trow = 4 * lg + (li >> 2);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
      taddr[dt] = (unsigned)(uintptr_t)smem + (unsigned)(trow * 128 + (((2 * dt + ((li & 3) >> 1)) ^ bwd_swz(trow)) & 7) * 16 + (li & 1) * 8);
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) faddr[kh] = smem + li * 128 + (((kh * 4 + lg) ^ bwd_swz(li)) & 7) * 16;
  }
  auto dma_issue = [&](unsigned char* st, int kv0) {
    const bf16_t* base = p.qkv + ((tok0 + kv0) * 3 * C + h * HD);
    unsigned char* dK = st + wave * 2048;
    unsigned char* dV = st + KTILE + wave * 2048;
    if (kv0 + 64 <= N) {
#pragma unroll
      for (int i = 0; i < 2; ++i) bwd_glds16(base + dma_off[i], dK + i * 1024);
#pragma unroll
      for (int i = 0; i < 2; ++i) bwd_glds16(base + dma_off[2 + i], dV + i * 1024);
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const bool ok = kv0 + dma_row[i] < N;
        bwd_glds16((const bf16_t*)(uintptr_t)(ok ? (uint64_t)(uintptr_t)(base + dma_off[i]) : zpage), dK + i * 1024);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const bool ok = kv0 + dma_row[i] < N;
        bwd_glds16((const bf16_t*)(uintptr_t)(ok ? (uint64_t)(uintptr_t)(base + dma_off[2 + i]) : zpage), dV + i * 1024);
      }
    }
  };

  f32x4 dq[2][4];
#pragma unroll
  for (int sub = 0; sub < 2; ++sub)
#pragma unroll
    for (int t = 0; t < 4; ++t) dq[sub][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float sc2 = p.scale * LOG2E, inv_scale = 1.0f / p.scale;
  const bool add_raw = p.drawlog != nullptr && p.T > 0 && qb == 0 && wave == 0 && li < p.T;

  const int nkv = (N + 63) / 64;
  if (VER == 2) {
    dma_issue(smem, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    stage_load(0);
    stage_store(smem);
  }
  __syncthreads();

  auto tile = [&](auto stage_tag, int j) {
    constexpr int ST = decltype(stage_tag)::value;   // -1: run-time stage (VER 0)
    const bool more = j + 1 < nkv;
    if (more) {
      if (VER == 2) dma_issue(smem + (1 - ST) * STAGE, (j + 1) * 64);
      else stage_load((j + 1) * 64);
    }
    const unsigned char* Kh = smem + (ST < 0 ? (j & 1) : ST) * STAGE;
    const unsigned char* Kt = Kh + KTILE;
    const unsigned char* Vh = Kh + 2 * KTILE;
    const int kv0 = j * 64;
    if (active) {
      const bool full = kv0 + 64 <= N;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {                 // two 32-key halves: keeps the live score registers at 2 x [2][2] tiles
        if (kv0 + 32 * ks >= N) continue;              // (block-uniform) nothing valid in this half of the last tile
        f32x4 s[2][2], dp[2][2];
        u32x2 ktl[4], kth[4];                          // VER 2: K^T fragments of this half (transpose reads)
        if constexpr (VER == 2) {
          u32x4 kf[2][2], vf[2][2];                    // every LDS read of the half in flight before the first MFMA
#pragma unroll
          for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
              kf[k2][kh] = *(const u32x4*)(faddr[kh] + ST * STAGE + k2 * 2048 + (ks ? 4096 : 0));
              vf[k2][kh] = *(const u32x4*)(faddr[kh] + ST * STAGE + KTILE + k2 * 2048 + (ks ? 4096 : 0));
            }
          if (ks == 0) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) { ktl[dt] = bwd_ds_read_tr16<ST * STAGE>(taddr[dt]); kth[dt] = bwd_ds_read_tr16<ST * STAGE + 16 * 128>(taddr[dt]); }
          } else {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) { ktl[dt] = bwd_ds_read_tr16<ST * STAGE + 32 * 128>(taddr[dt]); kth[dt] = bwd_ds_read_tr16<ST * STAGE + 48 * 128>(taddr[dt]); }
          }
          BWD_WAIT_TR8(ktl, kth);
          __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int k2 = 0; k2 < 2; ++k2) {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) { s[sub][k2] = (f32x4){0.f, 0.f, 0.f, 0.f}; dp[sub][k2] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
#pragma unroll
              for (int sub = 0; sub < 2; ++sub) {
                s[sub][k2] = mfma16(kf[k2][kh], qf[sub][kh], s[sub][k2]);
                dp[sub][k2] = mfma16(vf[k2][kh], gf[sub][kh], dp[sub][k2]);
              }
          }
          __builtin_amdgcn_s_setprio(0);
        } else {
        if (VER >= 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
          const int kt = 2 * ks + k2;
#pragma unroll
          for (int sub = 0; sub < 2; ++sub) { s[sub][k2] = (f32x4){0.f, 0.f, 0.f, 0.f}; dp[sub][k2] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
          for (int kh = 0; kh < 2; ++kh) {
            const u32x4 kf = *(const u32x4*)(Kh + lds_off(kt * 16 + li, kh * 4 + lg));
            const u32x4 vf = *(const u32x4*)(Vh + lds_off(kt * 16 + li, kh * 4 + lg));
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
              s[sub][k2] = mfma16(kf, qf[sub][kh], s[sub][k2]);
              dp[sub][k2] = mfma16(vf, gf[sub][kh], dp[sub][k2]);
            }
          }
        }
        if (VER >= 1) __builtin_amdgcn_s_setprio(0);
        }
        // s[sub][k2][r] = S[q = li][key = kv0 + 16 (2ks + k2) + 4 lg + r]
        u32x4 dsb[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
          float ds[2][4];
#pragma unroll
          for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float pv = __builtin_amdgcn_exp2f(fmaf(s[sub][k2][r], sc2, -lse2[sub]));
              ds[k2][r] = pv * (dp[sub][k2][r] - Dq[sub]);
            }
          if (!full) {
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (kv0 + (2 * ks + k2) * 16 + lg * 4 + r >= N) ds[k2][r] = 0.f;
          }
          if (sub == 0 && add_raw) {
            const float* rl = p.drawlog + (bh * p.T + li) * N;
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int key = kv0 + (2 * ks + k2) * 16 + lg * 4 + r;
                if (key < N) ds[k2][r] += rl[key] * inv_scale;
              }
          }
          dsb[sub] = (u32x4){pack2(ds[0][0], ds[0][1]), pack2(ds[0][2], ds[0][3]), pack2(ds[1][0], ds[1][1]), pack2(ds[1][2], ds[1][3])};
        }
        // dQ^T[d][q] += K^T[d][keys of this half] dS^T
        if (VER >= 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          u32x4 ka;
          if constexpr (VER == 2) ka = (u32x4){ktl[dt][0], ktl[dt][1], kth[dt][0], kth[dt][1]};
          else ka = perm_frag(Kt, dt * 16 + li, ks, lg);
          dq[0][dt] = mfma16(ka, dsb[0], dq[0][dt]);
          dq[1][dt] = mfma16(ka, dsb[1], dq[1][dt]);
        }
        if (VER >= 1) __builtin_amdgcn_s_setprio(0);
      }
    }
    if (more) {
      if (VER == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else stage_store(smem + (ST < 0 ? ((j + 1) & 1) : (1 - ST)) * STAGE);
    }
    __syncthreads();
  };
  if (VER >= 1) {
    for (int j = 0; j < nkv; j += 2) {
      tile(std::integral_constant<int, 0>{}, j);
      if (j + 1 < nkv) tile(std::integral_constant<int, 1>{}, j + 1);
    }
  } else {
    for (int j = 0; j < nkv; ++j) tile(std::integral_constant<int, -1>{}, j);
  }
  // dq[sub][dt][r] = dQ[q = li][d = 16 dt + 4 lg + r] / scale
#pragma unroll
  for (int sub = 0; sub < 2; ++sub) {
    const int qrow = q0 + sub * 16 + li;
    if (qrow >= N) continue;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
      *(u32x2*)(p.dqkv + (tok0 + qrow) * 3 * C + h * HD + dt * 16 + lg * 4) =
          (u32x2){pack2(dq[sub][dt][0] * p.scale, dq[sub][dt][1] * p.scale), pack2(dq[sub][dt][2] * p.scale, dq[sub][dt][3] * p.scale)};
  }
}

// --------------------------------------------------------------------------------------------------------
template <int VER>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(const BwdP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int STAGE = (VER == 2 ? 2 : 4) * KTILE;  // Q, Q^T, dO, dO^T (VER 2: Q, dO)
  const int nkb = (p.N + 127) / 128;
  const int wi = xcd_remap(blockIdx.x, gridDim.x);
  const int kb = wi % nkb, bh_ = wi / nkb;
  const int h = bh_ % p.nH, b = bh_ / p.nH;
  const int N = p.N, C = p.nH * HD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int64_t tok0 = (int64_t)b * N;
  const int64_t bh = (int64_t)b * p.nH + h;
  const int key0 = kb * 128 + wave * 32;
  const bool active = key0 < N;

  u32x4 kf[2][2], vf[2][2], dummy;
#pragma unroll
  for (int kt = 0; kt < 2; ++kt) {
    const int krow = key0 + kt * 16 + li;
    const bool ok = krow < N;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      Raw8<false> r;
      load8_raw<false>(p.qkv, (tok0 + krow) * 3 * C + C + h * HD + kh * 32 + lg * 8, ok, r);
      cvt8<false, false>(ok, r, kf[kt][kh], dummy);
      load8_raw<false>(p.qkv, (tok0 + krow) * 3 * C + 2 * C + h * HD + kh * 32 + lg * 8, ok, r);
      cvt8<false, false>(ok, r, vf[kt][kh], dummy);
    }
  }

  const bool isQ = tid < 128;
  const int kq = tid & 15, rb = (tid >> 4) & 7;
  Raw8<false> raw[4];
  unsigned okm = 0;
  auto stage_load = [&](int q0) {
    if (VER >= 1 && q0 + 64 <= N) {                   // block-uniform: a full tile needs no per-row selects
      okm = 0xfu;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t row = tok0 + q0 + kq * 4 + i;
        raw[i].r0 = isQ ? *(const u32x4*)(p.qkv + (row * 3 * C + h * HD + rb * 8)) : *(const u32x4*)(p.dout + (row * C + h * HD + rb * 8));
      }
      return;
    }
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
    if (VER >= 1 && okm == 0xfu) {
#pragma unroll
      for (int i = 0; i < 4; ++i) sh[i] = raw[i].r0;
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) cvt8<false, false>((okm >> i) & 1u, raw[i], sh[i], dummy);
    }
    unsigned char* base = isQ ? st : st + 2 * KTILE;
    store_units<true>(base, base + KTILE, sh, kq, rb);
  };

  // VER 2: wave w moves query rows [16 w, 16 w + 16) of the Q and of the dO tile as two 1-KiB pieces each
  int dma_off[4], dma_row[2];
  uint64_t zpage = 0;
  unsigned taddr[4];                                 // Q^T / dO^T transpose-read addresses per d tile
  const unsigned char* faddr[2];                     // row fragment addresses per kh
  if (VER == 2) {
    zpage = (uint64_t)(uintptr_t)g_bwd_zero_page;
    asm volatile("" : "+s"(zpage));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = wave * 16 + i * 8 + (lane >> 3), pc = lane & 7;
      dma_row[i] = r;
      dma_off[i] = r * 3 * C + ((pc ^ bwd_swz(r)) & 7) * 8;
      dma_off[2 + i] = r * C + ((pc ^ bwd_swz(r)) & 7) * 8;
    }
    const int trow = 4 * lg + (li >> 2);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
      taddr[dt] = (unsigned)(uintptr_t)smem + (unsigned)(trow * 128 + (((2 * dt + ((li & 3) >> 1)) ^ bwd_swz(trow)) & 7) * 16 + (li & 1) * 8);
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) faddr[kh] = smem + li * 128 + (((kh * 4 + lg) ^ bwd_swz(li)) & 7) * 16;
  }
  auto dma_issue = [&](unsigned char* st, int q0) {
    const bf16_t* baseQ = p.qkv + ((tok0 + q0) * 3 * C + h * HD);
    const bf16_t* baseG = p.dout + ((tok0 + q0) * C + h * HD);
    unsigned char* dQ = st + wave * 2048;
    unsigned char* dG = st + KTILE + wave * 2048;
    if (q0 + 64 <= N) {
#pragma unroll
      for (int i = 0; i < 2; ++i) bwd_glds16(baseQ + dma_off[i], dQ + i * 1024);
#pragma unroll
      for (int i = 0; i < 2; ++i) bwd_glds16(baseG + dma_off[2 + i], dG + i * 1024);
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const bool ok = q0 + dma_row[i] < N;
        bwd_glds16((const bf16_t*)(uintptr_t)(ok ? (uint64_t)(uintptr_t)(baseQ + dma_off[i]) : zpage), dQ + i * 1024);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const bool ok = q0 + dma_row[i] < N;
        bwd_glds16((const bf16_t*)(uintptr_t)(ok ? (uint64_t)(uintptr_t)(baseG + dma_off[2 + i]) : zpage), dG + i * 1024);
      }
    }
  };

  f32x4 dk[2][4], dv[2][4];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int t = 0; t < 4; ++t) { dk[kt][t] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[kt][t] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  const float sc2 = p.scale * LOG2E, inv_scale = 1.0f / p.scale;
  const float* Drow = p.stat + bh * 2 * p.Np;
  const float* Lrow = Drow + p.Np;

  const int nq = (N + 63) / 64;
  if (VER == 2) {
    dma_issue(smem, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    stage_load(0);
    stage_store(smem);
  }
  __syncthreads();

  auto tile = [&](auto stage_tag, int j) {
    constexpr int ST = decltype(stage_tag)::value;
    const bool more = j + 1 < nq;
    if (more) {
      if (VER == 2) dma_issue(smem + (1 - ST) * STAGE, (j + 1) * 64);
      else stage_load((j + 1) * 64);
    }
    const unsigned char* Qh = smem + (ST < 0 ? (j & 1) : ST) * STAGE;
    const unsigned char* Qt = Qh + KTILE;
    const unsigned char* Gh = Qh + 2 * KTILE;
    const unsigned char* Gt = Qh + 3 * KTILE;
    const int q0 = j * 64;
    if (active) {
      const bool full = q0 + 64 <= N;
      const bool add_raw = p.drawlog != nullptr && j == 0 && p.T > 0;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {                 // two 32-query halves
        if (q0 + 32 * ks >= N) continue;               // (block-uniform) nothing valid in this half of the last tile
        f32x4 s[2][2], dp[2][2];                       // [q sub of the half][key tile]
        float4 D4[2], L4[2];
        u32x2 qtl[4], qth[4], gtl[4], gth[4];          // VER 2: Q^T / dO^T fragments of this half (transpose reads)
        if constexpr (VER == 2) {
          u32x4 qa[2][2], ga[2][2];                    // every LDS read of the half in flight before the first MFMA
#pragma unroll
          for (int q2 = 0; q2 < 2; ++q2) {
            const int qr = q0 + (2 * ks + q2) * 16 + lg * 4;
            D4[q2] = qr < p.Np ? *(const float4*)(Drow + qr) : make_float4(0.f, 0.f, 0.f, 0.f);
            L4[q2] = qr < p.Np ? *(const float4*)(Lrow + qr) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
              qa[q2][kh] = *(const u32x4*)(faddr[kh] + ST * STAGE + q2 * 2048 + (ks ? 4096 : 0));
              ga[q2][kh] = *(const u32x4*)(faddr[kh] + ST * STAGE + KTILE + q2 * 2048 + (ks ? 4096 : 0));
            }
          }
          if (ks == 0) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
              qtl[dt] = bwd_ds_read_tr16<ST * STAGE>(taddr[dt]); qth[dt] = bwd_ds_read_tr16<ST * STAGE + 16 * 128>(taddr[dt]);
              gtl[dt] = bwd_ds_read_tr16<ST * STAGE + KTILE>(taddr[dt]); gth[dt] = bwd_ds_read_tr16<ST * STAGE + KTILE + 16 * 128>(taddr[dt]);
            }
          } else {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
              qtl[dt] = bwd_ds_read_tr16<ST * STAGE + 32 * 128>(taddr[dt]); qth[dt] = bwd_ds_read_tr16<ST * STAGE + 48 * 128>(taddr[dt]);
              gtl[dt] = bwd_ds_read_tr16<ST * STAGE + KTILE + 32 * 128>(taddr[dt]); gth[dt] = bwd_ds_read_tr16<ST * STAGE + KTILE + 48 * 128>(taddr[dt]);
            }
          }
          BWD_WAIT_TR16(qtl, qth, gtl, gth);
          __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int q2 = 0; q2 < 2; ++q2) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) { s[q2][kt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dp[q2][kt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
#pragma unroll
              for (int kt = 0; kt < 2; ++kt) {
                s[q2][kt] = mfma16(qa[q2][kh], kf[kt][kh], s[q2][kt]);
                dp[q2][kt] = mfma16(ga[q2][kh], vf[kt][kh], dp[q2][kt]);
              }
          }
          __builtin_amdgcn_s_setprio(0);
        } else {
        if (VER >= 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int q2 = 0; q2 < 2; ++q2) {
          const int qs = 2 * ks + q2;
          const int qr = q0 + qs * 16 + lg * 4;        // stat rows are padded to a multiple of 4 (zeros)
          D4[q2] = qr < p.Np ? *(const float4*)(Drow + qr) : make_float4(0.f, 0.f, 0.f, 0.f);
          L4[q2] = qr < p.Np ? *(const float4*)(Lrow + qr) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int kt = 0; kt < 2; ++kt) { s[q2][kt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dp[q2][kt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
          for (int kh = 0; kh < 2; ++kh) {
            const u32x4 qa = *(const u32x4*)(Qh + lds_off(qs * 16 + li, kh * 4 + lg));
            const u32x4 ga = *(const u32x4*)(Gh + lds_off(qs * 16 + li, kh * 4 + lg));
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
              s[q2][kt] = mfma16(qa, kf[kt][kh], s[q2][kt]);
              dp[q2][kt] = mfma16(ga, vf[kt][kh], dp[q2][kt]);
            }
          }
        }
        if (VER >= 1) __builtin_amdgcn_s_setprio(0);
        }
        // s[q2][kt][r] = S[q = q0 + 16 (2ks + q2) + 4 lg + r][key = key0 + 16 kt + li]
        u32x4 pb[2], dsb[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
          float pv[2][4], ds[2][4];
#pragma unroll
          for (int q2 = 0; q2 < 2; ++q2) {
            const float lr[4] = {L4[q2].x, L4[q2].y, L4[q2].z, L4[q2].w};
            const float dr[4] = {D4[q2].x, D4[q2].y, D4[q2].z, D4[q2].w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              pv[q2][r] = __builtin_amdgcn_exp2f(fmaf(s[q2][kt][r], sc2, -lr[r]));
              if (!full && q0 + (2 * ks + q2) * 16 + lg * 4 + r >= N) pv[q2][r] = 0.f;
              ds[q2][r] = pv[q2][r] * (dp[q2][kt][r] - dr[r]);
            }
          }
          if (add_raw && ks == 0) {
            const int key = key0 + kt * 16 + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int q = lg * 4 + r;                  // query sub 0 of the first tile
              if (q < p.T && key < N) ds[0][r] += p.drawlog[(bh * p.T + q) * N + key] * inv_scale;
            }
          }
          pb[kt] = (u32x4){pack2(pv[0][0], pv[0][1]), pack2(pv[0][2], pv[0][3]), pack2(pv[1][0], pv[1][1]), pack2(pv[1][2], pv[1][3])};
          dsb[kt] = (u32x4){pack2(ds[0][0], ds[0][1]), pack2(ds[0][2], ds[0][3]), pack2(ds[1][0], ds[1][1]), pack2(ds[1][2], ds[1][3])};
        }
        // dV^T[d][key] += dO^T[d][q half] P ;  dK^T[d][key] += Q^T[d][q half] dS
        if (VER >= 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          u32x4 ga, qa;
          if constexpr (VER == 2) {
            ga = (u32x4){gtl[dt][0], gtl[dt][1], gth[dt][0], gth[dt][1]};
            qa = (u32x4){qtl[dt][0], qtl[dt][1], qth[dt][0], qth[dt][1]};
          } else {
            ga = perm_frag(Gt, dt * 16 + li, ks, lg);
            qa = perm_frag(Qt, dt * 16 + li, ks, lg);
          }
#pragma unroll
          for (int kt = 0; kt < 2; ++kt) {
            dv[kt][dt] = mfma16(ga, pb[kt], dv[kt][dt]);
            dk[kt][dt] = mfma16(qa, dsb[kt], dk[kt][dt]);
          }
        }
        if (VER >= 1) __builtin_amdgcn_s_setprio(0);
      }
    }
    if (more) {
      if (VER == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else stage_store(smem + (ST < 0 ? ((j + 1) & 1) : (1 - ST)) * STAGE);
    }
    __syncthreads();
  };
  if (VER >= 1) {
    for (int j = 0; j < nq; j += 2) {
      tile(std::integral_constant<int, 0>{}, j);
      if (j + 1 < nq) tile(std::integral_constant<int, 1>{}, j + 1);
    }
  } else {
    for (int j = 0; j < nq; ++j) tile(std::integral_constant<int, -1>{}, j);
  }
  // dk[kt][dt][r] = dK[key = key0 + 16 kt + li][d = 16 dt + 4 lg + r] / scale
#pragma unroll
  for (int kt = 0; kt < 2; ++kt) {
    const int krow = key0 + kt * 16 + li;
    if (krow >= N) continue;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      bf16_t* dst = p.dqkv + (tok0 + krow) * 3 * C + C + h * HD + dt * 16 + lg * 4;
      *(u32x2*)dst = (u32x2){pack2(dk[kt][dt][0] * p.scale, dk[kt][dt][1] * p.scale), pack2(dk[kt][dt][2] * p.scale, dk[kt][dt][3] * p.scale)};
      *(u32x2*)(dst + C) = (u32x2){pack2(dv[kt][dt][0], dv[kt][dt][1]), pack2(dv[kt][dt][2], dv[kt][dt][3])};
    }
  }
}

}  // namespace

extern "C" int mtt_attn_bwd(const mtt_attn_desc* d, const void* dout, const float* drawlog, void* dqkv, float* stat, void* stream) {
  if (!d || !d->qkv || !d->out || !d->lse || !dout || !dqkv || !stat) return MTT_E_BADARG;
  if (d->B <= 0 || d->N <= 0 || d->nH <= 0 || d->T < 0 || d->T > 16) return MTT_E_BADARG;
  if (d->dtype != MTT_BF16 || d->prec != MTT_PREC_BF16) return MTT_E_UNSUPPORTED;
  if (((uintptr_t)d->qkv | (uintptr_t)dout | (uintptr_t)d->out | (uintptr_t)stat | (uintptr_t)dqkv) & 15) return MTT_E_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  constexpr int smem_dq = 2 * 3 * KTILE, smem_dkv = 2 * 4 * KTILE, smem_dma = 2 * 2 * KTILE;
  static std::atomic<unsigned long long> done_dq0{0}, done_dkv0{0}, done_dq1{0}, done_dkv1{0};
  const int ver = d->variant == MTT_ATTN_FAST_V0 ? 0 : d->variant == MTT_ATTN_FAST_V1 ? 1 : 2;
  if (ver == 0) {
    if (int e = mtt_ensure_dyn_lds((const void*)attn_bwd_dq_kernel<0>, smem_dq, done_dq0)) return e;
    if (int e = mtt_ensure_dyn_lds((const void*)attn_bwd_dkv_kernel<0>, smem_dkv, done_dkv0)) return e;
  } else if (ver == 1) {
    if (int e = mtt_ensure_dyn_lds((const void*)attn_bwd_dq_kernel<1>, smem_dq, done_dq1)) return e;
    if (int e = mtt_ensure_dyn_lds((const void*)attn_bwd_dkv_kernel<1>, smem_dkv, done_dkv1)) return e;
  }                                                  // VER 2 stays within the default 64 KiB of dynamic LDS
  const int Np = (d->N + 3) & ~3;
  const int64_t chunks = (int64_t)d->B * d->N * d->nH * 8;
  hipLaunchKernelGGL(attn_stat_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, s, (const bf16_t*)d->out, (const bf16_t*)dout,
                     d->lse, stat, d->B, d->N, d->nH, Np);
  BwdP p{(const bf16_t*)d->qkv, (const bf16_t*)dout, stat, d->T > 0 ? drawlog : nullptr, (bf16_t*)dqkv, d->B, d->N, d->nH, d->T, Np, d->scale};
  dim3 grid((unsigned)(((d->N + 127) / 128) * d->nH * d->B));
  if (ver == 0) {
    hipLaunchKernelGGL(attn_bwd_dq_kernel<0>, grid, dim3(256), smem_dq, s, p);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<0>, grid, dim3(256), smem_dkv, s, p);
  } else if (ver == 1) {
    hipLaunchKernelGGL(attn_bwd_dq_kernel<1>, grid, dim3(256), smem_dq, s, p);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<1>, grid, dim3(256), smem_dkv, s, p);
  } else {
    hipLaunchKernelGGL(attn_bwd_dq_kernel<2>, grid, dim3(256), smem_dma, s, p);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<2>, grid, dim3(256), smem_dma, s, p);
  }
  return (int)hipGetLastError();
}
