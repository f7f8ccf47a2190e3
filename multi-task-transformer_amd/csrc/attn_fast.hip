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

// Workgroup = 64 query rows (all 4 waves carry all 64 rows: 4 B fragments per K / V^T fragment read); the 128 keys of a
// staged tile are split over the waves (wave w owns keys 32w .. 32w+31): each wave runs an independent online softmax over
// its key subset (own running max / sum / O^T partial, like split-KV decoding) and the four partials are merged once, in
// LDS, at the end.  Compared with splitting the query rows over the waves this divides the LDS fragment traffic per MFMA by
// 3 (that variant was LDS-bandwidth bound: 24 KB of fragment reads per 32 MFMAs per wave).
__global__ __launch_bounds__(256, 2) void attn_fwd_fast_kernel(const AttnP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int STAGE = 4 * KTILE;                  // K [128 keys][64 d] (2 tiles), V^T 2 x [64 d][64 keys]
  const mtt_attn_desc& d = p.d;
  // 1-D grid, XCD-aware: the query blocks of one (batch, head) are consecutive work items of ONE XCD, so its K / V are fetched
  // into that XCD's L2 once and re-used by all of them (round-robin placement gave every query block a different L2: the
  // kernel was bound by the latency of L2-missing tile loads)
  const int nqb = (d.N + 63) / 64;
  const int wi = xcd_remap(blockIdx.x, gridDim.x);
  const int qb = wi % nqb, bh = wi / nqb;
  const int h = bh % d.nH, b = bh / d.nH;
  const int N = d.N, C = d.nH * HD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int64_t tok0 = (int64_t)b * N;
  const bf16_t* qkv = (const bf16_t*)d.qkv;
  const int q0 = qb * 64;

  u32x4 qf[4][2], dummy;
#pragma unroll
  for (int sub = 0; sub < 4; ++sub) {
    const int qrow = q0 + sub * 16 + li;
    const bool ok = qrow < N;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      Raw8<false> r;
      load8_raw<false>(qkv, (tok0 + qrow) * 3 * C + h * HD + kh * 32 + lg * 8, ok, r);
      cvt8<false, false>(ok, r, qf[sub][kh], dummy);
    }
  }

  // staging: every thread copies 4 K chunks and transposes one 4-key x 8-d unit of V
  Raw8<false> rawk[4], rawv[4];
  unsigned okk = 0, okv = 0;
  const int vq = tid & 31, vb = tid >> 5;            // V unit: keys 4 vq .. 4 vq + 3, d chunk vb
  auto stage_load = [&](int kv0) {
    okk = okv = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + 256 * i;                  // K chunk: key idx >> 3, d chunk idx & 7
      const int key = kv0 + (idx >> 3);
      const bool ok = key < N;
      okk |= (ok ? 1u : 0u) << i;
      load8_raw<false>(qkv, (tok0 + key) * 3 * C + C + h * HD + (idx & 7) * 8, ok, rawk[i]);
      const int vkey = kv0 + vq * 4 + i;
      const bool okv_ = vkey < N;
      okv |= (okv_ ? 1u : 0u) << i;
      load8_raw<false>(qkv, (tok0 + vkey) * 3 * C + 2 * C + h * HD + vb * 8, okv_, rawv[i]);
    }
  };
  auto stage_store = [&](unsigned char* st) {
    u32x4 sh[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + 256 * i;
      u32x4 kk;
      cvt8<false, false>((okk >> i) & 1u, rawk[i], kk, dummy);
      *(u32x4*)(st + lds_off(idx >> 3, idx & 7)) = kk;                 // rows 0..127: two consecutive 8 KiB tiles
      cvt8<false, false>((okv >> i) & 1u, rawv[i], sh[i], dummy);
    }
    u32x2 piece[8];
    transpose4x8(sh, piece);
    unsigned char* vt = st + 2 * KTILE + (vq >> 4) * KTILE;           // keys 0..63 -> first V^T tile, 64..127 -> second
    const int kq = vq & 15;
#pragma unroll
    for (int j = 0; j < 8; ++j) *(u32x2*)(vt + lds_off(vb * 8 + j, kq >> 1) + (kq & 1) * 8) = piece[j];
  };

  f32x4 o[4][4];
#pragma unroll
  for (int sub = 0; sub < 4; ++sub)
#pragma unroll
    for (int t = 0; t < 4; ++t) o[sub][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m_run[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, l_part[4] = {0.f, 0.f, 0.f, 0.f};
  const float sc2 = d.scale * LOG2E;
  const bool write_raw = d.rawlog != nullptr && d.T > 0 && qb == 0 && li < d.T;

  const int nkv = (N + 127) / 128;
  stage_load(0);
  stage_store(smem);
  __syncthreads();

  for (int j = 0; j < nkv; ++j) {
    const bool more = j + 1 < nkv;
    if (more) stage_load((j + 1) * 128);
    const unsigned char* Kh = smem + (j & 1) * STAGE;
    const unsigned char* Vt = Kh + 2 * KTILE + (wave >> 1) * KTILE;
    const int kbase = j * 128 + wave * 32;            // first key of this wave's 32-key slice
    if (kbase < N) {
      // ---- S^T = K Q^T : s[sub][kt][r] = S[q = 16 sub + li][key = kbase + 16 kt + 4 lg + r] ---------------------
      f32x4 s[4][2];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) s[sub][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
          const u32x4 kf = *(const u32x4*)(Kh + lds_off(wave * 32 + kt * 16 + li, kh * 4 + lg));
#pragma unroll
          for (int sub = 0; sub < 4; ++sub) s[sub][kt] = mfma16(kf, qf[sub][kh], s[sub][kt]);
        }
      }
      if (write_raw) {
        float* rl = d.rawlog + (((int64_t)b * d.nH + h) * d.T + li) * N;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = kbase + kt * 16 + lg * 4 + r;
            if (key < N) rl[key] = s[0][kt][r];
          }
      }
      // softmax in the log2 domain on the RAW scores: p = 2^(s*sc2 - m) is one fma + one v_exp per element; masking only in
      // the (wave-uniform) last slice; O^T is rescaled only when some lane's running max moved (alpha == 1 otherwise).
      const bool full = kbase + 32 <= N;
      if (!full) {
#pragma unroll
        for (int sub = 0; sub < 4; ++sub)
#pragma unroll
          for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if ((kbase + kt * 16 + lg * 4 + r) >= N) s[sub][kt][r] = -INFINITY;
      }
      u32x4 pb[4];
#pragma unroll
      for (int sub = 0; sub < 4; ++sub) {
        float mx = fmaxf(fmaxf(fmaxf(s[sub][0][0], s[sub][0][1]), fmaxf(s[sub][0][2], s[sub][0][3])),
                         fmaxf(fmaxf(s[sub][1][0], s[sub][1][1]), fmaxf(s[sub][1][2], s[sub][1][3])));
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
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pv = __builtin_amdgcn_exp2f(fmaf(s[sub][kt][r], sc2, -m_new));
            s[sub][kt][r] = pv;
            rs += pv;
          }
        l_part[sub] += rs;
        pb[sub] = (u32x4){pack2(s[sub][0][0], s[sub][0][1]), pack2(s[sub][0][2], s[sub][0][3]),
                          pack2(s[sub][1][0], s[sub][1][1]), pack2(s[sub][1][2], s[sub][1][3])};
      }
      // ---- O^T += V^T P^T over this wave's 32 keys -----------------------------------------------------------------
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const u32x4 vf = perm_frag(Vt, dt * 16 + li, wave & 1, lg);
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) o[sub][dt] = mfma16(vf, pb[sub], o[sub][dt]);
      }
    }
    if (more) stage_store(smem + ((j + 1) & 1) * STAGE);
    __syncthreads();
  }

  // ---- merge the four key-slice partials: O = sum_w e_w O_w / sum_w e_w l_w, e_w = 2^(m_w - m) -----------------------
  constexpr int PSTR = 68;                             // fp32 row pitch of the partials (64 + 4: rows start on different banks)
  float* stat = (float*)(smem + 4 * 64 * PSTR * 4);    // [2][4 waves][64 q]: m, l
  float* part = (float*)smem;                          // [4 waves][64 q][PSTR] fp32 (reuses the staging buffers)
#pragma unroll
  for (int sub = 0; sub < 4; ++sub) {
    float l = l_part[sub];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    l_part[sub] = l;
    if (lg == 0) { stat[wave * 64 + sub * 16 + li] = m_run[sub]; stat[256 + wave * 64 + sub * 16 + li] = l; }
  }
  __syncthreads();
#pragma unroll
  for (int sub = 0; sub < 4; ++sub) {
    const int q = sub * 16 + li;
    float m = fmaxf(fmaxf(stat[q], stat[64 + q]), fmaxf(stat[128 + q], stat[192 + q]));
    float l = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) l += __builtin_amdgcn_exp2f(stat[w * 64 + q] - m) * stat[256 + w * 64 + q];
    const float f = __builtin_amdgcn_exp2f(m_run[sub] - m) / l;        // m_run = -inf (no key seen) -> 0
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
      *(float4*)(part + ((wave * 64 + q) * PSTR + dt * 16 + lg * 4)) =
          make_float4(o[sub][dt][0] * f, o[sub][dt][1] * f, o[sub][dt][2] * f, o[sub][dt][3] * f);
  }
  __syncthreads();
  {
    const int q = tid >> 2, ds = (tid & 3) * 16;
    const int qrow = q0 + q;
    if (qrow < N) {
      float acc[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float4 v = *(const float4*)(part + ((w * 64 + q) * PSTR + ds + c * 4));
          acc[c * 4] += v.x; acc[c * 4 + 1] += v.y; acc[c * 4 + 2] += v.z; acc[c * 4 + 3] += v.w;
        }
      bf16_t* out = (bf16_t*)d.out + (tok0 + qrow) * C + h * HD + ds;
      *(u32x4*)out = (u32x4){pack2(acc[0], acc[1]), pack2(acc[2], acc[3]), pack2(acc[4], acc[5]), pack2(acc[6], acc[7])};
      *(u32x4*)(out + 8) = (u32x4){pack2(acc[8], acc[9]), pack2(acc[10], acc[11]), pack2(acc[12], acc[13]), pack2(acc[14], acc[15])};
      if (d.lse && (tid & 3) == 0) {
        const float m = fmaxf(fmaxf(stat[q], stat[64 + q]), fmaxf(stat[128 + q], stat[192 + q]));
        float l = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) l += __builtin_amdgcn_exp2f(stat[w * 64 + q] - m) * stat[256 + w * 64 + q];
        d.lse[((int64_t)b * d.nH + h) * N + qrow] = (m + log2f(l)) * 0.6931471805599453f;
      }
    }
  }
}

}  // namespace

// called by mtt_attn_fwd (attn.hip) for bf16 storage + MTT_PREC_BF16
int mtt_attn_fwd_fast(const mtt_attn_desc* dd, hipStream_t s) {
  constexpr int smem = 4 * 64 * 68 * 4 + 2 * 4 * 64 * 4;      // merge buffers (69.6 KB) >= 2 staging stages (64 KB)
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)attn_fwd_fast_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  AttnP p; p.d = *dd;
  dim3 grid((unsigned)(((dd->N + 63) / 64) * dd->nH * dd->B));
  hipLaunchKernelGGL(attn_fwd_fast_kernel, grid, dim3(256), smem, s, p);
  return (int)hipGetLastError();
}
