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

// ---- VER 2 staging: HBM -> LDS by LDS-DMA (no staging registers, no register transposes) --------------------------------------------
// K tile: [64 keys][8 chunks], the usual lds_off swizzle, applied to the SOURCE chunk of each DMA lane (the LDS image of a
// global_load_lds is lane-linear: lane l of a 1-KiB piece fills bytes 16 l .. 16 l + 15 = row l >> 3, position l & 7).
// V tile: [64 keys][4 units of 32 B] as it sits in memory (NOT transposed): the V^T fragments of O^T += V^T P^T come from
// ds_read_b64_tr_b16 (inside a 16-lane group, lane r pointing at row k0 + (r >> 2), columns c0 + 4 (r & 3) receives rows k0 .. k0 + 3 of
// column c0 + r: profiles/r02_probe_ds_read_b64_tr_b16.txt); 32-byte units are XOR-ed with (key >> 1) & 3 so that the 8 rows a 32-lane
// half touches land on 8 different bank octets.
__device__ __attribute__((aligned(32))) const unsigned g_attn_zero_page[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
MTT_DEV void attn_glds16(const bf16_t* src, unsigned char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
template <int OFF>
MTT_DEV u32x2 attn_ds_read_tr16(unsigned addr) {
  u32x2 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
  return r;
}
// Workgroup = 128 query rows, 32 per wave (two B fragments of Q per K / V^T fragment read); 64-key tiles.
// (A key-split variant — all waves share 64 query rows, each wave owns 32 keys of a 128-key tile, partials merged in LDS —
// measured 5 % slower on MI355X: the kernel is bound by VALU issue (softmax, staging address math), not by LDS traffic.)
// VER 2 (the default): K / V tiles go HBM -> LDS by LDS-DMA (no staging registers, no register transposes, no ds_write), the V^T fragments
// come from ds_read_b64_tr_b16, all 8 K fragment reads of a tile are in flight before its first MFMA, and the two cross-group reductions
// of the softmax use v_permlane16/32_swap instead of __shfl_xor (= ds_bpermute_b32 + a full LDS wait, 4 per tile on the critical path).
// 162 VGPRs -> 3 workgroups per CU.  B = 63, N = 1030, 16 heads: 605 -> 417 us per layer (656 TFLOP/s), N = 8194: 1 838 -> 1 306 us
// (842 TFLOP/s), outputs bitwise identical to VER 1 / VER 0 on every shape of tools/attn_variants_check.py (profiles/r02_attn_bench_w_*).
// VER 1 (mtt_attn_desc.variant = MTT_ATTN_FAST_V1): register-staged tiles (global loads -> 4x8 register transposes -> ds_write), KV loop
// unrolled by the two LDS stages, full tiles staged without per-key predicates, MFMA clusters at raised wave priority (650 -> 605 us over
// VER 0 = MTT_ATTN_FAST_V0).  Dead ends: forcing 3 / 4 workgroups per CU on VER 1 spills (806 / 1 669 us); softmax denominators on the
// matrix pipe (an A fragment of ones) changed nothing.
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

  // VER 2: wave w moves key rows [16 w, 16 w + 16) of the K and of the V tile as two 1-KiB pieces each
  int dma_off[4];                                    // element offsets from (token kv0 of this image, column 0 of this head): K piece 0/1, V piece 0/1
  int dma_row[2];
  uint64_t zpage = 0;
  if (VER >= 2) {
    zpage = (uint64_t)(uintptr_t)g_attn_zero_page;
    asm volatile("" : "+s"(zpage));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = wave * 16 + i * 8 + (lane >> 3), pc = lane & 7;
      dma_row[i] = r;
      dma_off[i] = r * 3 * C + C + ((pc ^ (r >> 1)) & 7) * 8;                  // K swizzle of VER 2: chunk ^ ((row >> 1) & 7), the same for every 16-key tile
      dma_off[2 + i] = r * 3 * C + 2 * C + (((((pc >> 1) ^ (r >> 1)) & 3) << 1) | (pc & 1)) * 8;
    }
  }
  auto dma_issue = [&](unsigned char* st, int kv0) {
    const bf16_t* base = qkv + ((tok0 + kv0) * 3 * C + h * HD);
    unsigned char* dK = st + wave * 2048;
    unsigned char* dV = st + KTILE + wave * 2048;
    if (kv0 + KV <= N) {
#pragma unroll
      for (int i = 0; i < 2; ++i) attn_glds16(base + dma_off[i], dK + i * 1024);
#pragma unroll
      for (int i = 0; i < 2; ++i) attn_glds16(base + dma_off[2 + i], dV + i * 1024);
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const bool ok = kv0 + dma_row[i] < N;
        attn_glds16((const bf16_t*)(uintptr_t)(ok ? (uint64_t)(uintptr_t)(base + dma_off[i]) : zpage), dK + i * 1024);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const bool ok = kv0 + dma_row[i] < N;
        attn_glds16((const bf16_t*)(uintptr_t)(ok ? (uint64_t)(uintptr_t)(base + dma_off[2 + i]) : zpage), dV + i * 1024);
      }
    }
  };
  // V^T fragment addresses (stage 0, ks = 0, dt = 0, first 4-key half): row 4 lg + (li >> 2), 8 bytes at column 4 (li & 3) of unit dt ^ f(row)
  const int vrow = 4 * lg + (li >> 2);
  const int vf_ = (vrow >> 1) & 3;                   // f(row); rows + 16 / + 32 / + 48 keep it
  unsigned vaddr[4];                                 // per d tile; stage, 32-key half and 4-key half are immediate offsets
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) vaddr[dt] = (unsigned)(uintptr_t)smem + (unsigned)(KTILE + vrow * 128 + (li & 3) * 8 + ((dt ^ vf_) & 3) * 32);
  // K fragment addresses of VER 2 (row 16 kt + li, chunk 4 kh + lg): two per-lane bases, the key tile and the stage are immediates
  const unsigned char* kaddr[2];
#pragma unroll
  for (int kh = 0; kh < 2; ++kh) kaddr[kh] = smem + li * 128 + (((kh * 4 + lg) ^ (li >> 1)) & 7) * 16;

  f32x4 o[2][4];
#pragma unroll
  for (int sub = 0; sub < 2; ++sub)
#pragma unroll
    for (int t = 0; t < 4; ++t) o[sub][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m_run[2] = {-INFINITY, -INFINITY}, l_part[2] = {0.f, 0.f};
  const float sc2 = d.scale * LOG2E;
  const bool write_raw = d.rawlog != nullptr && d.T > 0 && qb == 0 && wave == 0 && li < d.T;

  const int nkv = (N + KV - 1) / KV;
  if (VER >= 2) {
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
      if (VER >= 2) dma_issue(smem + (1 - ST) * STAGE, (j + 1) * KV);
      else stage_load((j + 1) * KV);
    }
    const unsigned char* Kh = smem + (ST < 0 ? (j & 1) : ST) * STAGE;
    const unsigned char* Vt = Kh + KTILE;
    const int kv0 = j * KV;
    if (active) {
      // ---- S^T = K Q^T : s[sub][kt][r] = S[q = li][key = kv0 + 16 kt + 4 lg + r] ------------------------------
      f32x4 s[2][4];
      if constexpr (VER >= 2) {
        u32x4 kf[4][2];                                // all 8 fragment reads in flight before the first MFMA
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int kh = 0; kh < 2; ++kh) kf[kt][kh] = *(const u32x4*)(kaddr[kh] + ST * STAGE + kt * 2048);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
          s[0][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
          s[1][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kh = 0; kh < 2; ++kh) {
            s[0][kt] = mfma16(kf[kt][kh], qf[0][kh], s[0][kt]);
            s[1][kt] = mfma16(kf[kt][kh], qf[1][kh], s[1][kt]);
          }
        }
        __builtin_amdgcn_s_setprio(0);
      } else {
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
      }
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
        if (VER >= 2) {
          mx = groups_max(mx);
        } else {
          mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
          mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        }
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
      if constexpr (VER >= 2) {
        // the 16 transpose reads of the tile are issued up front, unconditionally and with NO control flow before their waits: the asm loads
        // are invisible to the compiler's wait counting, so a register copy it places between a read and its wait (e.g. a phi move at a
        // branch merge) would copy a register still in flight.  The waits name every destination as an in/out operand: no use can move
        // above them.  Half 1 lands under half 0's MFMAs.
        const bool half1 = kv0 + 32 < N;               // block-uniform: the second 32 keys of the last tile may be all padding (P = 0, V = 0)
        u32x2 v0l[4], v0h[4], v1l[4], v1h[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          v0l[dt] = attn_ds_read_tr16<ST * STAGE>(vaddr[dt]);
          v0h[dt] = attn_ds_read_tr16<ST * STAGE + 16 * 128>(vaddr[dt]);
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          v1l[dt] = attn_ds_read_tr16<ST * STAGE + 32 * 128>(vaddr[dt]);
          v1h[dt] = attn_ds_read_tr16<ST * STAGE + 48 * 128>(vaddr[dt]);
        }
#define ATW(x) "+v"(x)
        asm volatile("s_waitcnt lgkmcnt(8)"
                     : ATW(v0l[0]), ATW(v0l[1]), ATW(v0l[2]), ATW(v0l[3]), ATW(v0h[0]), ATW(v0h[1]), ATW(v0h[2]), ATW(v0h[3]),
                       ATW(v1l[0]), ATW(v1l[1]), ATW(v1l[2]), ATW(v1l[3]), ATW(v1h[0]), ATW(v1h[1]), ATW(v1h[2]), ATW(v1h[3]) :: "memory");
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const u32x4 vf = (u32x4){v0l[dt][0], v0l[dt][1], v0h[dt][0], v0h[dt][1]};
          o[0][dt] = mfma16(vf, pb[0][0], o[0][dt]);
          o[1][dt] = mfma16(vf, pb[1][0], o[1][dt]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : ATW(v1l[0]), ATW(v1l[1]), ATW(v1l[2]), ATW(v1l[3]), ATW(v1h[0]), ATW(v1h[1]), ATW(v1h[2]), ATW(v1h[3]) :: "memory");
        if (half1) {
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
            const u32x4 vf = (u32x4){v1l[dt][0], v1l[dt][1], v1h[dt][0], v1h[dt][1]};
            o[0][dt] = mfma16(vf, pb[0][1], o[0][dt]);
            o[1][dt] = mfma16(vf, pb[1][1], o[1][dt]);
          }
        }
        __builtin_amdgcn_s_setprio(0);
#undef ATW
      } else {
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
    }
    if (more) {
      if (VER >= 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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

  // ---- epilogue: o[sub][dt][r] = O[q = li][d = 16 dt + 4 lg + r] ---------------------------------------------------
  bf16_t* out = (bf16_t*)d.out;
#pragma unroll
  for (int sub = 0; sub < 2; ++sub) {
    float l = l_part[sub];
    if (VER >= 2) {
      l = groups_sum(l);
    } else {
      l += __shfl_xor(l, 16, 64);
      l += __shfl_xor(l, 32, 64);
    }
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


// ---------------------------------------------------------------------------------------------------------------------------------
// attn_fwd_x3_kernel: the same swapped-product flash kernel in fp32-class arithmetic on MTT_SPLIT planes (x = hi + lo bf16; the forward
// of the x3f training mode and of x3f inference).  qkv arrives as two planes, every product is three MFMAs
//   S^T = Kh Qh^T + Kh Ql^T + Kl Qh^T,      O^T += Vh^T Ph^T + Vh^T Pl^T + Vl^T Ph^T      (P = Ph + Pl split in registers)
// and the output leaves as two planes again (the A operand of the LDS-DMA proj GEMM).  Staging as VER 2: the K / V tiles of BOTH planes
// go HBM -> LDS by LDS-DMA (4 x 8 KiB per stage, 2 stages = 64 KiB, 2 workgroups per CU); K fragments by ds_read_b128, V^T fragments
// by ds_read_b64_tr_b16.  The hi and the lo plane of an operand take turns in the SAME fragment registers (pass 1: Kh against Qh and Ql,
// pass 2: Kl against Qh; likewise V), so the kernel stays under 256 VGPRs.  Softmax statistics, raw prompt-row logits and the
// log-sum-exp (for the bf16 flash backward of the x3f mode) are fp32 exactly as in the bf16 kernel.
// Inline-asm transpose reads follow the rule of tools/check_tr_hazards.py: all reads of a group are issued unconditionally, the waits
// name every destination, no control flow in between.
// ---------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void attn_fwd_x3_kernel(const AttnP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int STAGE = 4 * KTILE;                  // K hi, K lo, V hi, V lo
  const mtt_attn_desc& d = p.d;
  const int nqb = (d.N + 127) / 128;
  const int wi = xcd_remap(blockIdx.x, gridDim.x);
  const int qb = wi % nqb, bh = wi / nqb;
  const int h = bh % d.nH, b = bh / d.nH;
  const int N = d.N, C = d.nH * HD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int64_t tok0 = (int64_t)b * N;
  const bf16_t* qh_ = (const bf16_t*)d.qkv;
  const bf16_t* ql_ = (const bf16_t*)d.qkv_lo;
  const int q0 = qb * 128 + wave * 32;
  const bool active = q0 < N;

  u32x4 qfh[2][2], qfl[2][2];
#pragma unroll
  for (int sub = 0; sub < 2; ++sub) {
    const int qrow = q0 + sub * 16 + li;
    const bool ok = qrow < N;
    const int64_t qi = (tok0 + (ok ? qrow : 0)) * 3 * C + h * HD + lg * 8;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      const u32x4 z = (u32x4){0u, 0u, 0u, 0u};
      qfh[sub][kh] = ok ? *(const u32x4*)(qh_ + qi + kh * 32) : z;
      qfl[sub][kh] = ok ? *(const u32x4*)(ql_ + qi + kh * 32) : z;
    }
  }

  // LDS-DMA: wave w moves key rows [16 w, 16 w + 16) of the K and of the V tile of both planes as two 1-KiB pieces each
  int dma_off[4], dma_row[2];
  uint64_t zpage = (uint64_t)(uintptr_t)g_attn_zero_page;
  asm volatile("" : "+s"(zpage));
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = wave * 16 + i * 8 + (lane >> 3), pc = lane & 7;
    dma_row[i] = r;
    dma_off[i] = r * 3 * C + C + ((pc ^ (r >> 1)) & 7) * 8;
    dma_off[2 + i] = r * 3 * C + 2 * C + (((((pc >> 1) ^ (r >> 1)) & 3) << 1) | (pc & 1)) * 8;
  }
  auto dma_issue = [&](unsigned char* st, int kv0) {
    const int64_t boff = (tok0 + kv0) * 3 * C + h * HD;
    const bool full = kv0 + KV <= N;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
      const bf16_t* base = (pl ? ql_ : qh_) + boff;
#pragma unroll
      for (int kv = 0; kv < 2; ++kv) {
        unsigned char* dst = st + (kv * 2 + pl) * KTILE + wave * 2048;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const bool ok = full || kv0 + dma_row[i] < N;
          attn_glds16((const bf16_t*)(uintptr_t)(ok ? (uint64_t)(uintptr_t)(base + dma_off[kv * 2 + i]) : zpage), dst + i * 1024);
        }
      }
    }
  };
  const int vrow = 4 * lg + (li >> 2);
  const int vf_ = (vrow >> 1) & 3;
  unsigned vaddr[4];                                 // V hi plane, stage 0; lo plane = + KTILE
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) vaddr[dt] = (unsigned)(uintptr_t)smem + (unsigned)(2 * KTILE + vrow * 128 + (li & 3) * 8 + ((dt ^ vf_) & 3) * 32);
  const unsigned char* kaddr[2];                     // K hi plane, stage 0; lo plane = + KTILE
#pragma unroll
  for (int kh = 0; kh < 2; ++kh) kaddr[kh] = smem + li * 128 + (((kh * 4 + lg) ^ (li >> 1)) & 7) * 16;

  f32x4 o[2][4];
#pragma unroll
  for (int sub = 0; sub < 2; ++sub)
#pragma unroll
    for (int t = 0; t < 4; ++t) o[sub][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m_run[2] = {-INFINITY, -INFINITY}, l_part[2] = {0.f, 0.f};
  const float sc2 = d.scale * LOG2E;
  const bool write_raw = d.rawlog != nullptr && d.T > 0 && qb == 0 && wave == 0 && li < d.T;

  const int nkv = (N + KV - 1) / KV;
  dma_issue(smem, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  auto tile = [&](auto stage_tag, int j) {
    constexpr int ST = decltype(stage_tag)::value;
    const bool more = j + 1 < nkv;
    if (more) dma_issue(smem + (1 - ST) * STAGE, (j + 1) * KV);
    const int kv0 = j * KV;
    if (active) {
      // ---- S^T = Kh Qh^T + Kh Ql^T + Kl Qh^T ----------------------------------------------------------------------------
      f32x4 s[2][4];
      {
        u32x4 kf[4][2];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int kh = 0; kh < 2; ++kh) kf[kt][kh] = *(const u32x4*)(kaddr[kh] + ST * STAGE + kt * 2048);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
          s[0][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
          s[1][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kh = 0; kh < 2; ++kh) {
            s[0][kt] = mfma16(kf[kt][kh], qfl[0][kh], s[0][kt]);
            s[1][kt] = mfma16(kf[kt][kh], qfl[1][kh], s[1][kt]);
            s[0][kt] = mfma16(kf[kt][kh], qfh[0][kh], s[0][kt]);
            s[1][kt] = mfma16(kf[kt][kh], qfh[1][kh], s[1][kt]);
          }
        }
        __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int kh = 0; kh < 2; ++kh) kf[kt][kh] = *(const u32x4*)(kaddr[kh] + ST * STAGE + KTILE + kt * 2048);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int kh = 0; kh < 2; ++kh) {
            s[0][kt] = mfma16(kf[kt][kh], qfh[0][kh], s[0][kt]);
            s[1][kt] = mfma16(kf[kt][kh], qfh[1][kh], s[1][kt]);
          }
        __builtin_amdgcn_s_setprio(0);
      }
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
      if (kv0 + KV > N) {
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
          for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if ((kv0 + kt * 16 + lg * 4 + r) >= N) s[sub][kt][r] = -INFINITY;
      }
      u32x4 pbh[2][2], pbl[2][2];
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {
        float mx = s[sub][0][0];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[sub][kt][r]);
        mx = groups_max(mx);
        const float m_new = fmaxf(m_run[sub], mx * sc2);
        if (__builtin_amdgcn_ballot_w64(m_new != m_run[sub]) != 0) {
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
        for (int ks = 0; ks < 2; ++ks) {
          const u32x4 ph = (u32x4){pack2(s[sub][2 * ks][0], s[sub][2 * ks][1]), pack2(s[sub][2 * ks][2], s[sub][2 * ks][3]),
                                   pack2(s[sub][2 * ks + 1][0], s[sub][2 * ks + 1][1]), pack2(s[sub][2 * ks + 1][2], s[sub][2 * ks + 1][3])};
          pbh[sub][ks] = ph;
          pbl[sub][ks] = (u32x4){pack2(s[sub][2 * ks][0] - lo_of(ph.x), s[sub][2 * ks][1] - hi_of(ph.x)),
                                 pack2(s[sub][2 * ks][2] - lo_of(ph.y), s[sub][2 * ks][3] - hi_of(ph.y)),
                                 pack2(s[sub][2 * ks + 1][0] - lo_of(ph.z), s[sub][2 * ks + 1][1] - hi_of(ph.z)),
                                 pack2(s[sub][2 * ks + 1][2] - lo_of(ph.w), s[sub][2 * ks + 1][3] - hi_of(ph.w))};
        }
      }
      // ---- O^T += Vh^T Ph^T + Vh^T Pl^T (pass 0), + Vl^T Ph^T (pass 1): the two planes take turns in the same fragment registers ----
      const bool half1 = kv0 + 32 < N;
#define ATW(x) "+v"(x)
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        u32x2 v0l[4], v0h[4], v1l[4], v1h[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          if (pass == 0) {
            v0l[dt] = attn_ds_read_tr16<ST * STAGE>(vaddr[dt]);
            v0h[dt] = attn_ds_read_tr16<ST * STAGE + 16 * 128>(vaddr[dt]);
          } else {
            v0l[dt] = attn_ds_read_tr16<ST * STAGE + KTILE>(vaddr[dt]);
            v0h[dt] = attn_ds_read_tr16<ST * STAGE + KTILE + 16 * 128>(vaddr[dt]);
          }
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          if (pass == 0) {
            v1l[dt] = attn_ds_read_tr16<ST * STAGE + 32 * 128>(vaddr[dt]);
            v1h[dt] = attn_ds_read_tr16<ST * STAGE + 48 * 128>(vaddr[dt]);
          } else {
            v1l[dt] = attn_ds_read_tr16<ST * STAGE + KTILE + 32 * 128>(vaddr[dt]);
            v1h[dt] = attn_ds_read_tr16<ST * STAGE + KTILE + 48 * 128>(vaddr[dt]);
          }
        }
        asm volatile("s_waitcnt lgkmcnt(8)"
                     : ATW(v0l[0]), ATW(v0l[1]), ATW(v0l[2]), ATW(v0l[3]), ATW(v0h[0]), ATW(v0h[1]), ATW(v0h[2]), ATW(v0h[3]),
                       ATW(v1l[0]), ATW(v1l[1]), ATW(v1l[2]), ATW(v1l[3]), ATW(v1h[0]), ATW(v1h[1]), ATW(v1h[2]), ATW(v1h[3]) :: "memory");
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const u32x4 vf = (u32x4){v0l[dt][0], v0l[dt][1], v0h[dt][0], v0h[dt][1]};
          if (pass == 0) {
            o[0][dt] = mfma16(vf, pbl[0][0], o[0][dt]);
            o[1][dt] = mfma16(vf, pbl[1][0], o[1][dt]);
          }
          o[0][dt] = mfma16(vf, pbh[0][0], o[0][dt]);
          o[1][dt] = mfma16(vf, pbh[1][0], o[1][dt]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : ATW(v1l[0]), ATW(v1l[1]), ATW(v1l[2]), ATW(v1l[3]), ATW(v1h[0]), ATW(v1h[1]), ATW(v1h[2]), ATW(v1h[3]) :: "memory");
        if (half1) {
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
            const u32x4 vf = (u32x4){v1l[dt][0], v1l[dt][1], v1h[dt][0], v1h[dt][1]};
            if (pass == 0) {
              o[0][dt] = mfma16(vf, pbl[0][1], o[0][dt]);
              o[1][dt] = mfma16(vf, pbl[1][1], o[1][dt]);
            }
            o[0][dt] = mfma16(vf, pbh[0][1], o[0][dt]);
            o[1][dt] = mfma16(vf, pbh[1][1], o[1][dt]);
          }
        }
        __builtin_amdgcn_s_setprio(0);
      }
#undef ATW
    }
    if (more) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };
  for (int j = 0; j < nkv; j += 2) {
    tile(std::integral_constant<int, 0>{}, j);
    if (j + 1 < nkv) tile(std::integral_constant<int, 1>{}, j + 1);
  }

  // ---- epilogue: o[sub][dt][r] = O[q = li][d = 16 dt + 4 lg + r], written as hi / lo planes ----------------------------------
  bf16_t* outh = (bf16_t*)d.out;
  bf16_t* outl = (bf16_t*)d.out_lo;
#pragma unroll
  for (int sub = 0; sub < 2; ++sub) {
    const float l = groups_sum(l_part[sub]);
    const int qrow = q0 + sub * 16 + li;
    if (qrow >= N) continue;
    const float inv = 1.0f / l;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const float v0 = o[sub][dt][0] * inv, v1 = o[sub][dt][1] * inv, v2 = o[sub][dt][2] * inv, v3 = o[sub][dt][3] * inv;
      const u32x2 hi = (u32x2){pack2(v0, v1), pack2(v2, v3)};
      const int64_t oi = (tok0 + qrow) * C + h * HD + dt * 16 + lg * 4;
      *(u32x2*)(outh + oi) = hi;
      *(u32x2*)(outl + oi) = (u32x2){pack2(v0 - lo_of(hi.x), v1 - hi_of(hi.x)), pack2(v2 - lo_of(hi.y), v3 - hi_of(hi.y))};
    }
    if (d.lse && lg == 0) d.lse[((int64_t)b * d.nH + h) * N + qrow] = (m_run[sub] + log2f(l)) * 0.6931471805599453f;
  }
}

}  // namespace

// called by mtt_attn_fwd (attn.hip) for MTT_SPLIT storage + MTT_PREC_X3
int mtt_attn_fwd_x3_split(const mtt_attn_desc* dd, hipStream_t s) {
  constexpr int smem = 2 * 4 * KTILE;
  static std::atomic<unsigned long long> done{0};
  if (int e = mtt_ensure_dyn_lds((const void*)attn_fwd_x3_kernel, smem, done)) return e;
  AttnP p; p.d = *dd;
  dim3 grid((unsigned)(((dd->N + 127) / 128) * dd->nH * dd->B));
  hipLaunchKernelGGL(attn_fwd_x3_kernel, grid, dim3(256), smem, s, p);
  return (int)hipGetLastError();
}

// called by mtt_attn_fwd (attn.hip) for bf16 storage + MTT_PREC_BF16
int mtt_attn_fwd_fast(const mtt_attn_desc* dd, hipStream_t s) {
  constexpr int smem = 2 * 2 * KTILE;
  AttnP p; p.d = *dd;
  dim3 grid((unsigned)(((dd->N + 127) / 128) * dd->nH * dd->B));
  if (dd->variant == MTT_ATTN_FAST_V0) hipLaunchKernelGGL(attn_fwd_fast_kernel<0>, grid, dim3(256), smem, s, p);
  else if (dd->variant == MTT_ATTN_FAST_V1) hipLaunchKernelGGL(attn_fwd_fast_kernel<1>, grid, dim3(256), smem, s, p);
  else hipLaunchKernelGGL(attn_fwd_fast_kernel<2>, grid, dim3(256), smem, s, p);
  return (int)hipGetLastError();
}
