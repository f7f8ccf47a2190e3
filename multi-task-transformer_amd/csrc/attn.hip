// mtt_attn_fwd: flash-style global attention over [T prompts || hw patches], head_dim 64, with the
// prompt-row raw-logit side channel (taskprompter.py:201-210; vit.py:184-191 with T = 0).
//
// One block = 64 query rows of one (batch, head): 4 waves x 16 rows.  K tiles ([64 keys][64 d]) and
// transposed V tiles ([64 d][64 keys], transposed in registers while staging) are double buffered in
// LDS and shared by the 4 waves; S = Q K^T and O += P V run on v_mfma_f32_16x16x32_bf16; the online
// softmax lives in registers (each lane owns 4 rows x 4 key columns per 16x16 tile; row reductions
// are 4 xor-shuffles inside a 16-lane group); P goes through a wave-private LDS tile to turn the
// MFMA C layout into an A fragment.  The N x N matrix never reaches HBM; the only extra output is
// rawlog[B, nH, T, N] (unscaled q.k of the T prompt rows) that cal_task_feature consumes.
#include <cstdlib>
#include "mtt_device.h"

namespace {

constexpr int KV = 64, HD = 64, QB = 64;
constexpr int KTILE = KV * HD * 2;   // 8 KiB per bf16 plane

struct AttnP { mtt_attn_desc d; };

// SRC: storage of qkv / out (compile time): 0 bf16, 1 fp32, 2 MTT_SPLIT (hi / lo bf16 planes: the x3 operands are read as stored, and
// the output is written as planes for the LDS-DMA proj GEMM); X3 implies SRC >= 1
template <int SRC> struct RawQ { Raw8<SRC == 1> a; u32x4 l; };
template <int SRC>
MTT_DEV void load_q8(const mtt_attn_desc& d, int64_t idx, bool ok, RawQ<SRC>& r) {
  load8_raw<SRC == 1>(d.qkv, idx, ok, r.a);
  if constexpr (SRC == 2) r.l = *(const u32x4*)((const bf16_t*)d.qkv_lo + (ok ? idx : 0));
}
template <bool X3, int SRC>
MTT_DEV void cvt_q8(bool ok, const RawQ<SRC>& r, u32x4& hi, u32x4& lo) {
  cvt8<X3 && SRC == 1, SRC == 1>(ok, r.a, hi, lo);
  if constexpr (SRC == 2) lo = ok ? r.l : (u32x4){0u, 0u, 0u, 0u};
}
template <bool X3, int SRC>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(const AttnP p) {
  constexpr bool F32 = SRC == 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NPL = X3 ? 2 : 1;
  constexpr int STAGE = KTILE * 2 * NPL;            // K planes, then Vt planes
  constexpr int PT = 16 * KV * 2;                   // 2 KiB per wave per plane
  unsigned char* const Pbase = smem + 2 * STAGE;

  const mtt_attn_desc& d = p.d;
  const int nqb = (d.N + QB - 1) / QB;                       // XCD-aware 1-D grid: one head's query blocks share an XCD's L2
  const int wi = xcd_remap(blockIdx.x, gridDim.x);
  const int qb = wi % nqb, bh = wi / nqb;
  const int h = bh % d.nH, b = bh / d.nH;
  const int N = d.N, C = d.nH * HD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int64_t tok0 = (int64_t)b * N;

  // ---- Q fragments (A operand), straight from global in fragment order ---------------------
  u32x4 qh[2], ql[2];
  {
    const int qrow = qb * QB + wave * 16 + li;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      RawQ<SRC> rq;
      load_q8<SRC>(d, (tok0 + qrow) * 3 * C + h * HD + kh * 32 + lg * 8, qrow < N, rq);
      cvt_q8<X3, SRC>(qrow < N, rq, qh[kh], ql[kh]);
    }
  }

  // ---- staging roles: waves 0,1 transpose V, waves 2,3 copy K --------------------------------
  const bool isV = tid < 128;
  RawQ<SRC> raw[4];
  unsigned okm = 0;
  const int kq = tid & 15, rb = (tid >> 4) & 7;     // V: 4 keys x 8 d unit
  const int kt_ = tid - 128;                        // K: chunk id base

  auto stage_load = [&](int kv0) {
    okm = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      // V: 4 keys x 8 d unit (transposed at store time);  K: 16-byte chunk (row = key, c = d chunk)
      const int idx = kt_ + 128 * i;
      const int key = isV ? kv0 + kq * 4 + i : kv0 + (idx >> 3);
      const int col = isV ? 2 * C + rb * 8 : C + (idx & 7) * 8;
      const bool ok = key < N;
      okm |= (ok ? 1u : 0u) << i;
      load_q8<SRC>(d, (tok0 + key) * 3 * C + col + h * HD, ok, raw[i]);
    }
  };
  auto stage_store = [&](unsigned char* st) {
    unsigned char* Kh = st;
    unsigned char* Vh = st + KTILE * NPL;
    u32x4 sh[4], sl[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) cvt_q8<X3, SRC>((okm >> i) & 1u, raw[i], sh[i], sl[i]);
    if (isV) {
      u32x2 piece[8];
      transpose4x8(sh, piece);
#pragma unroll
      for (int j = 0; j < 8; ++j) *(u32x2*)(Vh + lds_off(rb * 8 + j, kq >> 1) + (kq & 1) * 8) = piece[j];
      if (X3) {
        transpose4x8(sl, piece);
#pragma unroll
        for (int j = 0; j < 8; ++j) *(u32x2*)(Vh + KTILE + lds_off(rb * 8 + j, kq >> 1) + (kq & 1) * 8) = piece[j];
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int idx = kt_ + 128 * i, row = idx >> 3, c = idx & 7;
        *(u32x4*)(Kh + lds_off(row, c)) = sh[i];
        if (X3) *(u32x4*)(Kh + KTILE + lds_off(row, c)) = sl[i];
      }
    }
  };

  f32x4 o[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) o[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m_run[4], l_run[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { m_run[r] = -INFINITY; l_run[r] = 0.f; }
  const float sc2 = d.scale * 1.4426950408889634f;   // work in log2 domain
  const bool write_raw = d.rawlog != nullptr && d.T > 0 && qb == 0 && wave == 0;

  const int nkv = (N + KV - 1) / KV;
  stage_load(0);
  stage_store(smem);
  __syncthreads();

  unsigned char* const Pw = Pbase + wave * PT * NPL;

  for (int j = 0; j < nkv; ++j) {
    const bool more = j + 1 < nkv;
    if (more) stage_load((j + 1) * KV);
    const unsigned char* st = smem + (j & 1) * STAGE;
    const unsigned char* Kh = st;
    const unsigned char* Kl = st + KTILE;
    const unsigned char* Vh = st + KTILE * NPL;
    const unsigned char* Vl = Vh + KTILE;
    const int kv0 = j * KV;

    // ---- S = Q K^T (unscaled) ------------------------------------------------------------
    f32x4 s[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      s[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
        const u32x4 kf = *(const u32x4*)(Kh + lds_off(nt * 16 + li, kh * 4 + lg));
        if (X3) {
          const u32x4 kfl = *(const u32x4*)(Kl + lds_off(nt * 16 + li, kh * 4 + lg));
          s[nt] = mfma16(ql[kh], kf, s[nt]);
          s[nt] = mfma16(qh[kh], kfl, s[nt]);
        }
        s[nt] = mfma16(qh[kh], kf, s[nt]);
      }
    }
    // ---- prompt-row raw logits (side channel) ---------------------------------------------
    if (write_raw) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = lg * 4 + r;
        if (row < d.T) {
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) {
            const int key = kv0 + nt * 16 + li;
            if (key < N) d.rawlog[(((int64_t)b * d.nH + h) * d.T + row) * N + key] = s[nt][r];
          }
        }
      }
    }
    // ---- online softmax ----------------------------------------------------------------------
    float alpha[4];
    const bool full_tile = kv0 + KV <= N;            // block-uniform: only the last key tile needs masking
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float mx = -INFINITY;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        float t = s[nt][r] * sc2;
        if (!full_tile) t = (kv0 + nt * 16 + li) < N ? t : -INFINITY;
        s[nt][r] = t;
        mx = fmaxf(mx, t);
      }
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
      const float m_new = fmaxf(m_run[r], mx);
      alpha[r] = __builtin_amdgcn_exp2f(m_run[r] - m_new);
      float rs = 0.f;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const float pv = __builtin_amdgcn_exp2f(s[nt][r] - m_new);
        s[nt][r] = pv;
        rs += pv;
      }
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) rs += __shfl_xor(rs, off, 64);
      l_run[r] = l_run[r] * alpha[r] + rs;
      m_run[r] = m_new;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) o[t][r] *= alpha[r];

    // ---- P (C layout) -> wave-private LDS tile [16 rows][64 keys] -----------------------------
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = lg * 4 + r, col = nt * 16 + li;
        const int off = lds_off(row, col >> 3) + (col & 7) * 2;
        const bf16_t ph = f2bf(s[nt][r]);
        *(bf16_t*)(Pw + off) = ph;
        if (X3) *(bf16_t*)(Pw + PT + off) = f2bf(s[nt][r] - bf2f(ph));
      }
    // Pw is private to this wave and a wave's LDS operations execute in issue order: no block barrier needed,
    // only keep the compiler from moving the fragment reads above the stores.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- O += P V -----------------------------------------------------------------------------
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const u32x4 pf = *(const u32x4*)(Pw + lds_off(li, ks * 4 + lg));
      u32x4 pfl;
      if (X3) pfl = *(const u32x4*)(Pw + PT + lds_off(li, ks * 4 + lg));
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const u32x4 vf = *(const u32x4*)(Vh + lds_off(dt * 16 + li, ks * 4 + lg));
        if (X3) {
          const u32x4 vfl = *(const u32x4*)(Vl + lds_off(dt * 16 + li, ks * 4 + lg));
          o[dt] = mfma16(pfl, vf, o[dt]);
          o[dt] = mfma16(pf, vfl, o[dt]);
        }
        o[dt] = mfma16(pf, vf, o[dt]);
      }
    }

    if (more) stage_store(smem + ((j + 1) & 1) * STAGE);
    __syncthreads();
  }

  // ---- epilogue: normalise and store ------------------------------------------------------------
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int qrow = qb * QB + wave * 16 + lg * 4 + r;
    if (qrow >= N) continue;
    const float inv = 1.0f / l_run[r];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const int64_t oi = (tok0 + qrow) * C + h * HD + dt * 16 + li;
      const float ov = o[dt][r] * inv;
      if constexpr (SRC == 2) {
        const bf16_t hv = f2bf(ov);
        ((bf16_t*)d.out)[oi] = hv;
        ((bf16_t*)d.out_lo)[oi] = f2bf(ov - bf2f(hv));
      } else st_elem(d.out, oi, F32 ? MTT_F32 : MTT_BF16, ov);
    }
    if (d.lse && li == 0)
      d.lse[((int64_t)b * d.nH + h) * N + qrow] = (m_run[r] + log2f(l_run[r])) * 0.6931471805599453f;
  }
}

template <bool X3, int SRC>
int launch_attn(const AttnP& p, hipStream_t s) {
  constexpr int NPL = X3 ? 2 : 1;
  constexpr int smem = 2 * (KTILE * 2 * NPL) + 4 * (16 * KV * 2) * NPL;
  static std::atomic<unsigned long long> done{0};
  if (int e = mtt_ensure_dyn_lds((const void*)attn_fwd_kernel<X3, SRC>, smem, done)) return e;
  dim3 grid((unsigned)(((p.d.N + QB - 1) / QB) * p.d.nH * p.d.B));
  hipLaunchKernelGGL((attn_fwd_kernel<X3, SRC>), grid, dim3(256), smem, s, p);
  return (int)hipGetLastError();
}

}  // namespace

int mtt_attn_fwd_fast(const mtt_attn_desc* dd, hipStream_t s);     // attn_fast.hip
int mtt_attn_fwd_x3_split(const mtt_attn_desc* dd, hipStream_t s); // attn_fast.hip

extern "C" int mtt_attn_fwd(const mtt_attn_desc* dd, void* stream) {
  if (!dd || !dd->qkv || !dd->out) return MTT_E_BADARG;
  if (dd->B <= 0 || dd->N <= 0 || dd->nH <= 0 || dd->T < 0 || dd->T > 16) return MTT_E_BADARG;
  if (dd->prec == MTT_PREC_X3 && dd->dtype != MTT_F32 && dd->dtype != MTT_SPLIT) return MTT_E_UNSUPPORTED;
  if (dd->dtype == MTT_SPLIT && (dd->prec != MTT_PREC_X3 || !dd->qkv_lo || !dd->out_lo || ((uintptr_t)dd->qkv_lo & 15))) return MTT_E_BADARG;
  if ((uintptr_t)dd->qkv & 15) return MTT_E_ALIGN;
  if (dd->variant != MTT_ATTN_PLAIN && dd->prec == MTT_PREC_BF16 && dd->dtype == MTT_BF16 && !((uintptr_t)dd->out & 15))
    return mtt_attn_fwd_fast(dd, (hipStream_t)stream);
  AttnP p; p.d = *dd;
  if (dd->prec == MTT_PREC_X3 && dd->dtype == MTT_SPLIT && dd->variant != MTT_ATTN_PLAIN && !((uintptr_t)dd->out & 15) && !((uintptr_t)dd->out_lo & 15))
    return mtt_attn_fwd_x3_split(dd, (hipStream_t)stream);
  if (dd->prec == MTT_PREC_X3) return dd->dtype == MTT_SPLIT ? launch_attn<true, 2>(p, (hipStream_t)stream) : launch_attn<true, 1>(p, (hipStream_t)stream);
  return dd->dtype == MTT_F32 ? launch_attn<false, 1>(p, (hipStream_t)stream) : launch_attn<false, 0>(p, (hipStream_t)stream);
}
