// bf16 attention kernels in "swapped product" form (head_dim 64): every score tile is computed TRANSPOSED,
// S^T = K Q^T, so that a lane owns ONE query column (q = lane & 15) and 4 consecutive keys per 16-key tile.  Then
//   * softmax statistics are per-lane scalars (row max: 2 xor-shuffles across the 4 lane groups; row sums stay per-lane
//     partials until the epilogue),
//   * the MFMA C layout of P^T (keys 4g..4g+3 of two 16-key tiles) IS a valid B fragment of the next product
//     O^T = V^T P^T once the reduction index is permuted consistently on the A side (two 8-byte LDS reads of the V^T row
//     instead of one 16-byte read) — P never goes through LDS,
//   * each wave carries 32 query rows (two B fragments per K / V^T fragment read): 128 query rows per workgroup share one
//     staged K / V^T tile.
// The fp32-parity (x3) mode keeps the straightforward kernel in attn.hip.
#include "mtt_device.h"
#include <type_traits>

namespace {

constexpr int KV = 64, HD = 64;
constexpr int KTILE = KV * HD * 2;   // 8 KiB per bf16 tile
constexpr float LOG2E = 1.4426950408889634f;

struct AttnP { mtt_attn_desc d; };

// A fragment of a [rows][64] tile whose reduction index follows the C-layout permutation of a 32-wide k step:
// slots 0..3 <- cols 32*ks + 4g .. +3, slots 4..7 <- cols 32*ks + 16 + 4g .. +3
MTT_DEV u32x4 perm_frag(const unsigned char* tile, int row, int ks, int lg) {
  const u32x2 lo = *(const u32x2*)(tile + lds_off(row, 4 * ks + (lg >> 1)) + (lg & 1) * 8);
  const u32x2 hi = *(const u32x2*)(tile + lds_off(row, 4 * ks + 2 + (lg >> 1)) + (lg & 1) * 8);
  return (u32x4){lo[0], lo[1], hi[0], hi[1]};
}

// Workgroup = 128 query rows, 32 per wave (two B fragments of Q per K / V^T fragment read); 64-key tiles.
// (A key-split variant — all waves share 64 query rows, each wave owns 32 keys of a 128-key tile, partials merged in LDS —
// measured 5 % slower on MI355X: the kernel is bound by VALU issue (softmax, staging address math), not by LDS traffic.)
// VER 1 (the default; VER 0 = the previous form, mtt_attn_desc.variant = MTT_ATTN_FAST_V0, kept for A/B): the KV loop is unrolled by the two
// LDS stages (compile-time stage offsets: the fragment reads take immediate offsets instead of a per-read address add), full key tiles
// are staged without per-key predicates (the `key < N` selects only run for the sequence's last tile), and the MFMA clusters run at
// raised wave priority: 650 -> 605 us per layer at B = 63, N = 1030 and 2 004 -> 1 844 us at N = 8194, bitwise identical outputs
// (tools/attn_bench.py).  Asking the compiler for 3 / 4 workgroups per CU (<= 168 / 128 VGPRs) instead of 2 spills and is much slower
// (806 / 1 669 us), and taking the softmax denominators out of the matrix pipe (an A fragment of ones) changed nothing: measured, dropped.
template <int VER>
__global__ __launch_bounds__(256, 2) void attn_fwd_fast_kernel(const AttnP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int STAGE = 2 * KTILE;                  // K, V^T
  const mtt_attn_desc& d = p.d;
  // 1-D grid, XCD-aware: the query blocks of one (batch, head) are consecutive work items of one XCD (shared L2)
  const int nqb = (d.N + 127) / 128;
  const int wi = xcd_remap(blockIdx.x, gridDim.x);
  const int qb = wi % nqb, bh = wi / nqb;
  const int h = bh % d.nH, b = bh / d.nH;
  const int N = d.N, C = d.nH * HD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int64_t tok0 = (int64_t)b * N;
  const bf16_t* qkv = (const bf16_t*)d.qkv;
  const int q0 = qb * 128 + wave * 32;
  const bool active = q0 < N;

  u32x4 qf[2][2], dummy;
#pragma unroll
  for (int sub = 0; sub < 2; ++sub) {
    const int qrow = q0 + sub * 16 + li;
    const bool ok = qrow < N;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      Raw8<false> r;
      load8_raw<false>(qkv, (tok0 + qrow) * 3 * C + h * HD + kh * 32 + lg * 8, ok, r);
      cvt8<false, false>(ok, r, qf[sub][kh], dummy);
    }
  }

  // staging roles: waves 0,1 transpose V (4 keys x 8 d units), waves 2,3 copy K
  const bool isV = tid < 128;
  Raw8<false> raw[4];
  unsigned okm = 0;
  const int kq = tid & 15, rb = (tid >> 4) & 7;
  const int kt_ = tid - 128;
  auto stage_load = [&](int kv0) {
    if (VER >= 1 && kv0 + KV <= N) {                 // block-uniform: a full tile needs no per-key selects
      okm = 0xfu;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int idx = kt_ + 128 * i;
        const int key = isV ? kv0 + kq * 4 + i : kv0 + (idx >> 3);
        const int col = isV ? 2 * C + rb * 8 : C + (idx & 7) * 8;
        raw[i].r0 = *(const u32x4*)(qkv + ((tok0 + key) * 3 * C + col + h * HD));
      }
      return;
    }
    okm = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = kt_ + 128 * i;
      const int key = isV ? kv0 + kq * 4 + i : kv0 + (idx >> 3);
      const int col = isV ? 2 * C + rb * 8 : C + (idx & 7) * 8;
      const bool ok = key < N;
      okm |= (ok ? 1u : 0u) << i;
      load8_raw<false>(qkv, (tok0 + key) * 3 * C + col + h * HD, ok, raw[i]);
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
    if (isV) {
      u32x2 piece[8];
      transpose4x8(sh, piece);
#pragma unroll
      for (int j = 0; j < 8; ++j) *(u32x2*)(st + KTILE + lds_off(rb * 8 + j, kq >> 1) + (kq & 1) * 8) = piece[j];
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int idx = kt_ + 128 * i;
        *(u32x4*)(st + lds_off(idx >> 3, idx & 7)) = sh[i];
      }
    }
  };

  f32x4 o[2][4];
#pragma unroll
  for (int sub = 0; sub < 2; ++sub)
#pragma unroll
    for (int t = 0; t < 4; ++t) o[sub][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m_run[2] = {-INFINITY, -INFINITY}, l_part[2] = {0.f, 0.f};
  const float sc2 = d.scale * LOG2E;
  const bool write_raw = d.rawlog != nullptr && d.T > 0 && qb == 0 && wave == 0 && li < d.T;

  const int nkv = (N + KV - 1) / KV;
  stage_load(0);
  stage_store(smem);
  __syncthreads();

  auto tile = [&](auto stage_tag, int j) {
    constexpr int ST = decltype(stage_tag)::value;   // -1: run-time stage (VER 0)
    const bool more = j + 1 < nkv;
    if (more) stage_load((j + 1) * KV);
    const unsigned char* Kh = smem + (ST < 0 ? (j & 1) : ST) * STAGE;
    const unsigned char* Vt = Kh + KTILE;
    const int kv0 = j * KV;
    if (active) {
      // ---- S^T = K Q^T : s[sub][kt][r] = S[q = li][key = kv0 + 16 kt + 4 lg + r] ------------------------------
      f32x4 s[2][4];
      if (VER >= 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        s[0][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        s[1][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
          const u32x4 kf = *(const u32x4*)(Kh + lds_off(kt * 16 + li, kh * 4 + lg));
          s[0][kt] = mfma16(kf, qf[0][kh], s[0][kt]);
          s[1][kt] = mfma16(kf, qf[1][kh], s[1][kt]);
        }
      }
      if (VER >= 1) __builtin_amdgcn_s_setprio(0);
      if (write_raw) {
        float* rl = d.rawlog + (((int64_t)b * d.nH + h) * d.T + li) * N;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = kv0 + kt * 16 + lg * 4 + r;
            if (key < N) rl[key] = s[0][kt][r];
          }
      }
      // softmax in the log2 domain on the RAW scores: p = 2^(s*sc2 - m) is one fma + one v_exp per element; masking only in
      // the (block-uniform) last tile; O^T is rescaled only when some lane's running max moved (alpha == 1 otherwise).
      if (kv0 + KV > N) {
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
          for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if ((kv0 + kt * 16 + lg * 4 + r) >= N) s[sub][kt][r] = -INFINITY;
      }
      u32x4 pb[2][2];
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {
        float mx = s[sub][0][0];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[sub][kt][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run[sub], mx * sc2);
        if (__builtin_amdgcn_ballot_w64(m_new != m_run[sub]) != 0) {      // wave-uniform: rescale only when a max moved
          const float alpha = __builtin_amdgcn_exp2f(m_run[sub] - m_new);
          l_part[sub] *= alpha;
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[sub][t][r] *= alpha;
          m_run[sub] = m_new;
        }
        float rs = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pv = __builtin_amdgcn_exp2f(fmaf(s[sub][kt][r], sc2, -m_new));
            s[sub][kt][r] = pv;
            rs += pv;
          }
        l_part[sub] += rs;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
          pb[sub][ks] = (u32x4){pack2(s[sub][2 * ks][0], s[sub][2 * ks][1]), pack2(s[sub][2 * ks][2], s[sub][2 * ks][3]),
                                pack2(s[sub][2 * ks + 1][0], s[sub][2 * ks + 1][1]), pack2(s[sub][2 * ks + 1][2], s[sub][2 * ks + 1][3])};
      }
      // ---- O^T += V^T P^T ---------------------------------------------------------------------------------------------
      if (VER >= 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if (kv0 + 32 * ks >= N) continue;              // block-uniform: nothing valid in this half of the last tile (P = 0)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const u32x4 vf = perm_frag(Vt, dt * 16 + li, ks, lg);
          o[0][dt] = mfma16(vf, pb[0][ks], o[0][dt]);
          o[1][dt] = mfma16(vf, pb[1][ks], o[1][dt]);
        }
      }
      if (VER >= 1) __builtin_amdgcn_s_setprio(0);
    }
    if (more) stage_store(smem + (ST < 0 ? ((j + 1) & 1) : (1 - ST)) * STAGE);
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

  // ---- epilogue: o[sub][dt][r] = O[q = li][d = 16 dt + 4 lg + r] ---------------------------------------------------
  bf16_t* out = (bf16_t*)d.out;
#pragma unroll
  for (int sub = 0; sub < 2; ++sub) {
    float l = l_part[sub];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const int qrow = q0 + sub * 16 + li;
    if (qrow >= N) continue;
    const float inv = 1.0f / l;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
      *(u32x2*)(out + (tok0 + qrow) * C + h * HD + dt * 16 + lg * 4) =
          (u32x2){pack2(o[sub][dt][0] * inv, o[sub][dt][1] * inv), pack2(o[sub][dt][2] * inv, o[sub][dt][3] * inv)};
    if (d.lse && lg == 0) d.lse[((int64_t)b * d.nH + h) * N + qrow] = (m_run[sub] + log2f(l)) * 0.6931471805599453f;
  }
}

}  // namespace

// called by mtt_attn_fwd (attn.hip) for bf16 storage + MTT_PREC_BF16
int mtt_attn_fwd_fast(const mtt_attn_desc* dd, hipStream_t s) {
  constexpr int smem = 2 * 2 * KTILE;
  AttnP p; p.d = *dd;
  dim3 grid((unsigned)(((dd->N + 127) / 128) * dd->nH * dd->B));
  if (dd->variant == MTT_ATTN_FAST_V0) hipLaunchKernelGGL(attn_fwd_fast_kernel<0>, grid, dim3(256), smem, s, p);
  else hipLaunchKernelGGL(attn_fwd_fast_kernel<1>, grid, dim3(256), smem, s, p);
  return (int)hipGetLastError();
}
