// mtt_gemm: the one MFMA contraction kernel behind every Linear / 1x1 / 3x3 / ConvTranspose(k=s=2)
// matmul of the hot path and their dgrad / wgrad (see include/mtt_hip.h for the contract).
//
// Structure (round 1, correctness first): 128 x 128 x 64 block tile, 256 threads = 4 waves in a
// 2 x 2 grid, each wave 64 x 64 = 4 x 4 tiles of v_mfma_f32_16x16x32_bf16 (64 fp32 accumulators per
// lane).  Operands are register-staged global -> LDS (the stagers apply dtype conversion, the X3
// hi/lo split, the im2col gather and, for transposed operands, an in-register 4x8 transpose), LDS is
// double buffered with one barrier per K step, fragment reads are swizzled conflict-free
// ds_read_b128.  Blocks are remapped so consecutive tiles of one XCD share A rows in its L2.
#include "mtt_device.h"
#include <type_traits>

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;   // 16 KiB per bf16 plane

struct GemmP {
  mtt_gemm_desc d;
  FastDiv divW, divH, divCp, div3;   // conv geometry
  FastDiv divAmb, divDmb, divRmb;    // row-group mappings
  FastDiv divPsW, divPsH, divPsCo;   // pixel-shuffle store
  int tiles_m, tiles_n;
  int group_m;
};

MTT_DEV int64_t row_off(uint32_t m, int mb, int64_t bs, int64_t ld, FastDiv f) {
  if (mb <= 0) return (int64_t)m * ld;
  const uint32_t q = fdiv(m, f);
  return (int64_t)q * bs + (int64_t)(m - q * (uint32_t)mb) * ld;
}

// ---------------------------------------------------------------------------------------------
// Stagers: global -> registers (load) -> LDS tile (store).  All 256 threads take part.
// ---------------------------------------------------------------------------------------------

// 32 zero bytes: invalid chunks (row / k out of range, conv halo) are READ from here instead of being masked after the
// load — the register stagers are VALU-issue bound (every VALU instruction costs 4 SIMD cycles), and a pointer select is 2
// v_cndmask where masking the 16 loaded bytes (+ keeping the address in range) was 6.
__device__ __attribute__((aligned(32))) const unsigned g_zero_page[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};

// reduction index contiguous in memory.  4 chunks of 8 elements per thread.
template <bool X3, bool CONV, bool F32>
struct StagerK {
  const void* base; int dtype; int K; int c;
  int64_t roff[4]; bool rok[4];
  uint64_t zpage;                   // address of g_zero_page, kept opaque in SGPRs (otherwise re-materialised per load)
  unsigned tapmask[4];              // CONV: bit t set <=> tap t of this row reads inside the image
  int tap, ci, knext;               // CONV: incremental (tap, channel) of this thread's chunk; k the next load() expects
  Raw8<F32> raw[4];

  MTT_DEV void init(const GemmP& p, const void* b, int dt, int row0, int rows, int64_t ld, int mb, int64_t bs, FastDiv fmb) {
    base = b; dtype = dt; K = p.d.K; c = threadIdx.x & 7;
    zpage = (uint64_t)(uintptr_t)g_zero_page;
    asm volatile("" : "+s"(zpage));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = row0 + (threadIdx.x >> 3) + 32 * i;
      rok[i] = r < rows;
      const uint32_t rr = rok[i] ? (uint32_t)r : 0u;
      if (CONV) {
        const uint32_t t = fdiv(rr, p.divW);
        const int px = (int)(rr - t * (uint32_t)p.d.conv.W);
        const uint32_t bb = fdiv(t, p.divH);
        const int py = (int)(t - bb * (uint32_t)p.d.conv.H);
        roff[i] = (int64_t)rr * ld;
        unsigned m = 0;
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {
          int ty = tp / 3, tx = tp % 3;
          if (p.d.conv.flip) { ty = 2 - ty; tx = 2 - tx; }
          const int yy = py + (ty - 1) * p.d.conv.dil, xx = px + (tx - 1) * p.d.conv.dil;
          if (rok[i] && yy >= 0 && yy < p.d.conv.H && xx >= 0 && xx < p.d.conv.W) m |= 1u << tp;
        }
        tapmask[i] = m;
      } else {
        roff[i] = row_off(rr, mb, bs, ld, fmb);
        tapmask[i] = rok[i] ? 1u : 0u;
      }
    }
    ldx = ld;
    tap = 0; ci = c * 8; knext = 0;
    if (CONV) {                       // normalise (Cp may be smaller than 64)
      while (ci >= p.d.conv.Cp) { ci -= p.d.conv.Cp; ++tap; }
    }
  }
  int64_t ldx;

  MTT_DEV void load(const GemmP& p, int k0) {
    const int k = k0 + c * 8;
    bool kok = k < K;
    int64_t koff = k;
    unsigned bit = 0;
    if (CONV) {
      if (k0 != knext) {              // (never taken by the k loop: steps are consecutive)
        const uint32_t t = fdiv((uint32_t)(kok ? k : 0), p.divCp);
        tap = (int)t; ci = k - (int)t * p.d.conv.Cp;
      }
      kok = kok && ci < p.d.conv.C;
      int ty = (int)fdiv((uint32_t)tap, p.div3), tx = tap - 3 * ty;
      if (p.d.conv.flip) { ty = 2 - ty; tx = 2 - tx; }
      koff = (int64_t)((ty - 1) * p.d.conv.dil * p.d.conv.W + (tx - 1) * p.d.conv.dil) * ldx + ci;
      bit = (unsigned)tap;
      // advance to the next 64-wide step
      ci += 64; knext = k0 + 64;
      while (ci >= p.d.conv.Cp) { ci -= p.d.conv.Cp; ++tap; }
    }
    const int es = F32 ? 4 : 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool ok = kok && ((tapmask[i] >> bit) & 1u);
      const uint64_t real = (uint64_t)(uintptr_t)base + (uint64_t)((roff[i] + koff) * es);
      const unsigned char* ptr = (const unsigned char*)(uintptr_t)(ok ? real : zpage);
      if constexpr (F32) {
        raw[i].v0 = *(const float4*)ptr;
        raw[i].v1 = *(const float4*)(ptr + 16);
      } else {
        raw[i].r0 = *(const u32x4*)ptr;
      }
    }
  }

  MTT_DEV void store(unsigned char* t_hi, unsigned char* t_lo) const {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (threadIdx.x >> 3) + 32 * i;
      u32x4 hi, lo;
      cvt8<X3, F32>(true, raw[i], hi, lo);
      *(u32x4*)(t_hi + lds_off(row, c)) = hi;
      if (X3) *(u32x4*)(t_lo + lds_off(row, c)) = lo;
    }
  }
};

// row index contiguous in memory (transposed view): element(r, k) at base + k*ld + r.
// Each thread owns a 4 (k) x 8 (rows) unit, transposes it in registers, writes 8 x 8-byte pieces.
// CONV: B operand of the 3x3 wgrad — k = pixel, r = (tap, ci): base + pixel_shifted*ld + ci.
template <bool X3, bool CONV, bool F32>
struct StagerR {
  const void* base; int dtype; int K; int kq, rb;
  int64_t roff; bool rok; int64_t ld;
  int dy, dx;
  Raw8<F32> raw[4]; unsigned okm;

  MTT_DEV void init(const GemmP& p, const void* b, int dt, int row0, int rows, int64_t ld_) {
    base = b; dtype = dt; K = p.d.K; ld = ld_;
    kq = threadIdx.x & 15; rb = threadIdx.x >> 4;
    const int r = row0 + rb * 8;
    rok = r < rows;
    roff = r; dy = dx = 0;
    if (CONV) {
      const uint32_t rr = rok ? (uint32_t)r : 0u;
      const uint32_t tap = fdiv(rr, p.divCp);
      const int ci = (int)rr - (int)tap * p.d.conv.Cp;
      rok = rok && ci < p.d.conv.C;
      const int ty = (int)fdiv(tap, p.div3), tx = (int)tap - 3 * ty;
      dy = (ty - 1) * p.d.conv.dil; dx = (tx - 1) * p.d.conv.dil;
      roff = (int64_t)(dy * p.d.conv.W + dx) * ld + ci;
    }
  }

  MTT_DEV void load(const GemmP& p, int k0) {
    okm = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = k0 + kq * 4 + i;
      bool ok = rok && k < K;
      if (CONV) {
        const uint32_t kk = ok ? (uint32_t)k : 0u;
        const uint32_t t = fdiv(kk, p.divW);
        const int x = (int)(kk - t * (uint32_t)p.d.conv.W);
        const uint32_t bb = fdiv(t, p.divH);
        const int y = (int)(t - bb * (uint32_t)p.d.conv.H);
        const int yy = y + dy, xx = x + dx;
        ok = ok && yy >= 0 && yy < p.d.conv.H && xx >= 0 && xx < p.d.conv.W;
      }
      okm |= (ok ? 1u : 0u) << i;
      load8_raw<F32>(base, (int64_t)k * ld + roff, ok, raw[i]);
    }
  }

  MTT_DEV void store(unsigned char* t_hi, unsigned char* t_lo) const {
    u32x4 hi[4], lo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) cvt8<X3, F32>((okm >> i) & 1u, raw[i], hi[i], lo[i]);
    u32x2 piece[8];
    transpose4x8(hi, piece);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int row = rb * 8 + j;
      *(u32x2*)(t_hi + lds_off(row, kq >> 1) + (kq & 1) * 8) = piece[j];
    }
    if (X3) {
      transpose4x8(lo, piece);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int row = rb * 8 + j;
        *(u32x2*)(t_lo + lds_off(row, kq >> 1) + (kq & 1) * 8) = piece[j];
      }
    }
  }
};

template <int OP, bool X3, bool F32> struct StagerSel;
template <bool X3, bool F32> struct StagerSel<MTT_OP_K, X3, F32> { typedef StagerK<X3, false, F32> type; };
template <bool X3, bool F32> struct StagerSel<MTT_OP_CONV_K, X3, F32> { typedef StagerK<X3, true, F32> type; };
template <bool X3, bool F32> struct StagerSel<MTT_OP_R, X3, F32> { typedef StagerR<X3, false, F32> type; };
template <bool X3, bool F32> struct StagerSel<MTT_OP_CONV_R, X3, F32> { typedef StagerR<X3, true, F32> type; };

template <int OP, bool X3, typename S>
MTT_DEV void stager_init_a(S& s, const GemmP& p, const void* base, int row0) {
  if constexpr (OP == MTT_OP_K || OP == MTT_OP_CONV_K)
    s.init(p, base, p.d.a_dtype, row0, p.d.M, p.d.lda, p.d.a_mb, p.d.a_bs, p.divAmb);
  else
    s.init(p, base, p.d.a_dtype, row0, p.d.M, p.d.lda);
}
template <int OP, bool X3, typename S>
MTT_DEV void stager_init_b(S& s, const GemmP& p, const void* base, int row0) {
  if constexpr (OP == MTT_OP_K || OP == MTT_OP_CONV_K)
    s.init(p, base, p.d.b_dtype, row0, p.d.N, p.d.ldb, 0, 0, p.divAmb);
  else
    s.init(p, base, p.d.b_dtype, row0, p.d.N, p.d.ldb);
}

// ---------------------------------------------------------------------------------------------
// Shared epilogue.  Block tile TBN columns wide, WAVES_M x WAVES_N waves, each wave MT x NTL tiles of 16 x 16:
//   acc[a][b][r] = D[(wm*MT + a)*16 + lg*4 + r][(wn*NTL + b)*16 + li]
// ---------------------------------------------------------------------------------------------
// EPI_ABL (measurement only): 1 = everything but the global stores of the vector path, 2 = no LDS staging / barriers (stores of
// register garbage at the right addresses).
// SWP: the accumulators hold the TRANSPOSED MFMA result (acc[a][b][r] = D[(wm*MT + a)*16 + li][(wn*NTL + b)*16 + lg*4 + r], the layout of
// gemm_dma_kernel's direct-store variant); only the staging writes differ.
template <int TBN, int WAVES_M, int WAVES_N, int MT, int NTL, int EPI_ABL = 0, bool SWP = false>
MTT_DEV void gemm_epilogue(const GemmP& p, f32x4 (&acc)[MT][NTL], unsigned char* smem, int m0, int n0, int zo, int zi) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  constexpr int NTHREADS = 64 * WAVES_M * WAVES_N, CHUNKS = TBN / 8, RPP = NTHREADS / CHUNKS, NSLAB = WAVES_M * MT / 4;
  // ------------------------------- epilogue --------------------------------------------------
  // Accumulators go through LDS (two 64-row halves, [64][132] fp32) so that every thread owns 8 consecutive
  // columns of a row: column constants are loaded once, row addressing once per 8 outputs, and D / resid / aux
  // are moved with coalesced 16/32-byte accesses.
  const mtt_gemm_desc& d = p.d;
  const int64_t zcol = (int64_t)zo * d.col_zo + (int64_t)zi * d.col_zi;
  const int64_t zD = (int64_t)zo * d.d_zo + (int64_t)zi * d.d_zi;
  const int64_t zAux = (int64_t)zo * d.aux_zo + (int64_t)zi * d.aux_zi;
  const int64_t zR = (int64_t)zo * d.r_zo + (int64_t)zi * d.r_zi;
  const int n_store = d.n_store > d.N ? d.n_store : d.N;
  constexpr int EP_LD = TBN + 4;
  float* const ep = (float*)smem;

  const int c8 = threadIdx.x % CHUNKS;             // this thread's 8-column chunk of the tile
  const int ncol0 = n0 + c8 * 8;
  float cs[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int n = ncol0 + j;
    const bool nv = n < d.N;
    cs[j] = (d.colscale && nv) ? d.colscale[zcol + n] : 1.0f;
    sh[j] = (d.colshift && nv) ? d.colshift[zcol + n] : 0.0f;
  }
  const bool full_chunk = ncol0 + 8 <= d.N && d.store_mode == MTT_STORE_ROWS;

#pragma unroll 1
  for (int half = 0; half < NSLAB; ++half) {          // 64-row slabs of the block tile
#pragma unroll
    for (int a = 0; a < MT; ++a) {
      const int gt = wm * MT + a;                      // m-tile index inside the block tile
      if ((gt >> 2) != half) continue;
#pragma unroll
      for (int b = 0; b < NTL; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (EPI_ABL == 2) asm volatile("" :: "v"(acc[a][b][r]));
          else if (SWP) ep[((gt & 3) * 16 + li) * EP_LD + (wn * NTL + b) * 16 + lg * 4 + r] = acc[a][b][r];
          else ep[((gt & 3) * 16 + lg * 4 + r) * EP_LD + (wn * NTL + b) * 16 + li] = acc[a][b][r];
        }
    }
    if (EPI_ABL != 2) __syncthreads();
    if (ncol0 < n_store || (d.store_mode == MTT_STORE_PIXSHUF2 && ncol0 < d.N)) {
#pragma unroll 1
      for (int i = 0; i < 64 / RPP; ++i) {
        const int rl = threadIdx.x / CHUNKS + RPP * i;
        const int m = m0 + half * 64 + rl;
        if (m >= d.M) continue;
        float v[8];
        if (EPI_ABL == 2) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = (float)(rl + j);
        } else {
          const float4 lo4 = *(const float4*)(ep + rl * EP_LD + c8 * 8);
          const float4 hi4 = *(const float4*)(ep + rl * EP_LD + c8 * 8 + 4);
          v[0] = lo4.x; v[1] = lo4.y; v[2] = lo4.z; v[3] = lo4.w; v[4] = hi4.x; v[5] = hi4.y; v[6] = hi4.z; v[7] = hi4.w;
        }
        uint32_t q = 0, rem = (uint32_t)m;
        if (d.d_mb > 0) { q = fdiv((uint32_t)m, p.divDmb); rem = m - q * d.d_mb; }
        const int64_t doff = zD + (d.d_mb > 0 ? (int64_t)q * d.d_bs + (int64_t)rem * d.ldd : (int64_t)m * d.ldd) + ncol0;
        const int64_t auxoff = zAux + (int64_t)m * d.ldaux + ncol0;
        const float rs = d.rowscale ? d.rowscale[q * 2 + (rem >= (uint32_t)d.n_prompt ? 1 : 0)] : 1.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = v[j] * d.alpha * cs[j] + sh[j];
        if (full_chunk) {
          if (d.aux_out) {
            if (d.aux_dtype == MTT_F32) {
              *(float4*)((float*)d.aux_out + auxoff) = make_float4(v[0], v[1], v[2], v[3]);
              *(float4*)((float*)d.aux_out + auxoff + 4) = make_float4(v[4], v[5], v[6], v[7]);
            } else {
              *(u32x4*)((bf16_t*)d.aux_out + auxoff) = (u32x4){pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
            }
          }
          if (d.act == MTT_ACT_GELU) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = gelu_f(v[j]);
          } else if (d.act == MTT_ACT_RELU) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.0f);
          } else if (d.act == MTT_ACT_GELU_BWD || d.act == MTT_ACT_RELU_BWD) {
            float z[8];
            if (d.aux_dtype == MTT_F32) {
              const float4 z0 = *(const float4*)((const float*)d.aux_in + auxoff);
              const float4 z1 = *(const float4*)((const float*)d.aux_in + auxoff + 4);
              z[0] = z0.x; z[1] = z0.y; z[2] = z0.z; z[3] = z0.w; z[4] = z1.x; z[5] = z1.y; z[6] = z1.z; z[7] = z1.w;
            } else {
              const u32x4 u = *(const u32x4*)((const bf16_t*)d.aux_in + auxoff);
#pragma unroll
              for (int t = 0; t < 4; ++t) { z[2 * t] = lo_of(u[t]); z[2 * t + 1] = hi_of(u[t]); }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = d.act == MTT_ACT_GELU_BWD ? v[j] * gelu_grad_f(z[j]) : (z[j] > 0.0f ? v[j] : 0.0f);
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] *= rs;
          if (d.resid) {
            const int64_t roff = zR + row_off((uint32_t)m, d.r_mb, d.r_bs, d.ldr, p.divRmb) + ncol0;
            const float4 r0 = *(const float4*)(d.resid + roff);
            const float4 r1 = *(const float4*)(d.resid + roff + 4);
            v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w; v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
          }
          if (EPI_ABL == 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("" :: "v"(v[j]));
          } else if (d.d_dtype == MTT_F32) {
            *(float4*)((float*)d.D + doff) = make_float4(v[0], v[1], v[2], v[3]);
            *(float4*)((float*)d.D + doff + 4) = make_float4(v[4], v[5], v[6], v[7]);
          } else {
            *(u32x4*)((bf16_t*)d.D + doff) = (u32x4){pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
          }
        } else {
          // ragged chunk (N tail / channel padding) or pixel-shuffle store: element by element
          int64_t roff = 0;
          if (d.resid) roff = zR + row_off((uint32_t)m, d.r_mb, d.r_bs, d.ldr, p.divRmb) + ncol0;
#pragma unroll 1
          for (int j = 0; j < 8; ++j) {
            const int n = ncol0 + j;
            float w = 0.0f;
            if (n < d.N) {
              w = v[j];
              if (d.aux_out) st_elem(d.aux_out, auxoff + j, d.aux_dtype, w);
              if (d.act == MTT_ACT_GELU) w = gelu_f(w);
              else if (d.act == MTT_ACT_RELU) w = fmaxf(w, 0.0f);
              else if (d.act == MTT_ACT_GELU_BWD) w *= gelu_grad_f(ld_elem(d.aux_in, auxoff + j, d.aux_dtype));
              else if (d.act == MTT_ACT_RELU_BWD) w = ld_elem(d.aux_in, auxoff + j, d.aux_dtype) > 0.0f ? w : 0.0f;
              w *= rs;
              if (d.resid) w += d.resid[roff + j];
            }
            if (d.store_mode == MTT_STORE_PIXSHUF2) {
              if (n >= d.N) continue;
              const uint32_t qq = fdiv((uint32_t)n, p.divPsCo);
              const int co = n - (int)qq * d.ps_Co;
              const uint32_t t = fdiv((uint32_t)m, p.divPsW);
              const int x = m - (int)t * d.ps_W;
              const uint32_t bb = fdiv(t, p.divPsH);
              const int y = (int)t - (int)bb * d.ps_H;
              const int64_t orow = ((int64_t)bb * (2 * d.ps_H) + 2 * y + (int)(qq >> 1)) * (2 * d.ps_W) + 2 * x + (int)(qq & 1);
              st_elem(d.D, zD + orow * d.ldd + co, d.d_dtype, w);
            } else if (n < n_store) {
              st_elem(d.D, doff + j, d.d_dtype, w);
            }
          }
        }
      }
    }
    if (EPI_ABL != 2) __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// Specialised epilogue for INTERIOR tiles of the hot call sites.  The general epilogue above decides everything per row at run
// time (dtype, activation, aux, residual, row groups, ragged columns: ~150 instructions and a dozen scalar branches per 8 outputs);
// measured on MI355X it costs 12-18 us of a 40-46 us K = 1024 tile and is bound by INSTRUCTION ISSUE, not by its stores or its LDS
// staging (profiles/r02_gemm_ablate_g_epilogue_parts.log: removing the global stores saves 1.7 us, removing the staging 0.9 us).
// Here the feature set is a template parameter, the tile is known to be full (no bounds checks) and rows are contiguous, so a row of 8
// outputs is a handful of instructions.  KIND:
//   (acc below = acc * colscale + colshift; colscale is the folded eval-mode BatchNorm of the conv call sites, absent elsewhere)
//   0  D bf16 = acc + bias                              (qkv, bf16 dgrads)
//   1  D f32  = acc + bias                              (weight-gradient slabs, fp32 dgrads)
//   2  D bf16 = GELU(z), z = acc + bias; aux_out bf16 = z when given   (fc1)
//   3  D f32  = rowscale * (acc + bias) + resid         (proj / fc2 into the fp32 residual stream; rowscale / resid optional)
//   4  D bf16 = (acc + bias) * GELU'(aux_in bf16)       (fc2 dgrad)
// ---------------------------------------------------------------------------------------------
// feature set -> KIND (-1: general epilogue only); a pure function of the descriptor, shared by the device dispatch and the host policy
__host__ __device__ inline int epilogue_kind_of(const mtt_gemm_desc& d) {
  if (d.store_mode != MTT_STORE_ROWS || d.alpha != 1.0f) return -1;
  if (d.d_mb > 0 && d.d_bs != (int64_t)d.d_mb * d.ldd) return -1;                    // D rows contiguous
  const bool auxi = d.aux_in != nullptr, auxo = d.aux_out != nullptr;
  if ((auxi || auxo) && d.aux_dtype != MTT_BF16) return -1;
  if (d.resid && ((d.r_mb > 0 && d.r_bs != (int64_t)d.r_mb * d.ldr) || d.d_dtype != MTT_F32)) return -1;
  if (d.rowscale && (d.d_dtype != MTT_F32 || d.d_mb <= 0)) return -1;
  if (d.act == MTT_ACT_NONE && !auxi && !auxo) {
    if (d.d_dtype == MTT_F32) return (d.resid || d.rowscale) ? 3 : 1;
    return (d.resid || d.rowscale) ? -1 : 0;
  }
  if (d.act == MTT_ACT_GELU && d.d_dtype == MTT_BF16 && !d.resid && !d.rowscale && !auxi) return 2;
  if (d.act == MTT_ACT_GELU_BWD && d.d_dtype == MTT_BF16 && !d.resid && !d.rowscale && auxi && !auxo) return 4;
  return -1;
}
MTT_DEV int fast_epilogue_kind(const mtt_gemm_desc& d, int m0, int n0, int tbm, int tbn) {
  if (m0 + tbm > d.M || n0 + tbn > d.N) return -1;                                   // interior tiles only
  return epilogue_kind_of(d);
}

template <int KIND, int TBN, int WAVES_M, int WAVES_N, int MT, int NTL>
MTT_DEV void gemm_epilogue_fast(const GemmP& p, f32x4 (&acc)[MT][NTL], unsigned char* smem, int m0, int n0, int zo, int zi) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  constexpr int NTHREADS = 64 * WAVES_M * WAVES_N, CHUNKS = TBN / 8, RPP = NTHREADS / CHUNKS, NSLAB = WAVES_M * MT / 4, NIT = 64 / RPP;
  constexpr int EP_LD = TBN + 4;
  const mtt_gemm_desc& d = p.d;
  float* const ep = (float*)smem;
  const int c8 = threadIdx.x % CHUNKS, rl0 = threadIdx.x / CHUNKS;
  const int ncol0 = n0 + c8 * 8;
  float sh[8], cs[8];
  {
    const int64_t zcol = (int64_t)zo * d.col_zo + (int64_t)zi * d.col_zi + ncol0;
#pragma unroll
    for (int j = 0; j < 8; ++j) { sh[j] = d.colshift ? d.colshift[zcol + j] : 0.0f; cs[j] = d.colscale ? d.colscale[zcol + j] : 1.0f; }
  }
  const int64_t zD = (int64_t)zo * d.d_zo + (int64_t)zi * d.d_zi + ncol0;
  const int64_t zAux = (int64_t)zo * d.aux_zo + (int64_t)zi * d.aux_zi + ncol0;
  const int64_t zR = (int64_t)zo * d.r_zo + (int64_t)zi * d.r_zi + ncol0;
  const float* const epr = ep + rl0 * EP_LD + c8 * 8;

#pragma unroll 1
  for (int half = 0; half < NSLAB; ++half) {
#pragma unroll
    for (int a = 0; a < MT; ++a) {
      const int gt = wm * MT + a;
      if ((gt >> 2) != half) continue;
#pragma unroll
      for (int b = 0; b < NTL; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          ep[((gt & 3) * 16 + lg * 4 + r) * EP_LD + (wn * NTL + b) * 16 + li] = acc[a][b][r];
    }
    __syncthreads();
    const int mrow = m0 + half * 64 + rl0;                    // first of this thread's NIT rows (stride RPP)
    // KIND 3 / 4: this slab's residual / GELU' input rows, all issued before they are used
    float4 ra[NIT], rb[NIT];
    u32x4 za[NIT];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int64_t m = mrow + RPP * i;
      if (KIND == 3) {
        if (d.resid) { ra[i] = *(const float4*)(d.resid + (zR + m * d.ldr)); rb[i] = *(const float4*)(d.resid + (zR + m * d.ldr) + 4); }
        else { ra[i] = rb[i] = make_float4(0.f, 0.f, 0.f, 0.f); }
      }
      if (KIND == 4) za[i] = *(const u32x4*)((const bf16_t*)d.aux_in + (zAux + m * d.ldaux));
    }
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int64_t m = mrow + RPP * i;
      const float4 lo4 = *(const float4*)(epr + RPP * i * EP_LD);
      const float4 hi4 = *(const float4*)(epr + RPP * i * EP_LD + 4);
      float v[8] = {fmaf(lo4.x, cs[0], sh[0]), fmaf(lo4.y, cs[1], sh[1]), fmaf(lo4.z, cs[2], sh[2]), fmaf(lo4.w, cs[3], sh[3]),
                    fmaf(hi4.x, cs[4], sh[4]), fmaf(hi4.y, cs[5], sh[5]), fmaf(hi4.z, cs[6], sh[6]), fmaf(hi4.w, cs[7], sh[7])};
      if (KIND == 2) {
        if (d.aux_out)
          *(u32x4*)((bf16_t*)d.aux_out + (zAux + m * d.ldaux)) = (u32x4){pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = gelu_f(v[j]);
      }
      if (KIND == 4) {
        const u32x4 u = za[i];
        v[0] *= gelu_grad_f(lo_of(u.x)); v[1] *= gelu_grad_f(hi_of(u.x)); v[2] *= gelu_grad_f(lo_of(u.y)); v[3] *= gelu_grad_f(hi_of(u.y));
        v[4] *= gelu_grad_f(lo_of(u.z)); v[5] *= gelu_grad_f(hi_of(u.z)); v[6] *= gelu_grad_f(lo_of(u.w)); v[7] *= gelu_grad_f(hi_of(u.w));
      }
      if (KIND == 3) {
        if (d.rowscale) {
          const uint32_t q = fdiv((uint32_t)m, p.divDmb), rem = (uint32_t)m - q * (uint32_t)d.d_mb;
          const float rs = d.rowscale[q * 2 + (rem >= (uint32_t)d.n_prompt ? 1 : 0)];
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] *= rs;
        }
        v[0] += ra[i].x; v[1] += ra[i].y; v[2] += ra[i].z; v[3] += ra[i].w; v[4] += rb[i].x; v[5] += rb[i].y; v[6] += rb[i].z; v[7] += rb[i].w;
      }
      if (KIND == 1 || KIND == 3) {
        float* dp = (float*)d.D + (zD + m * d.ldd);
        *(float4*)dp = make_float4(v[0], v[1], v[2], v[3]);
        *(float4*)(dp + 4) = make_float4(v[4], v[5], v[6], v[7]);
      } else {
        *(u32x4*)((bf16_t*)d.D + (zD + m * d.ldd)) = (u32x4){pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// Direct-store epilogue for the 256 x 256 / 8-wave kernels when the MFMAs were issued with SWAPPED operands (acc = B-fragment x
// A-fragment): a lane then holds 4 consecutive COLUMNS of one row per 16 x 16 fragment, and one v_permlane16_swap per register between
// the fragments b and b + 1 (lane rows 0 / 2 keep fragment b and receive the neighbouring row's 4 columns, rows 1 / 3 likewise for
// b + 1; mapping measured in profiles/r02_probe_permlane_swap.txt) gives every lane 8 consecutive columns: exactly the per-thread work
// item of gemm_epilogue_fast, without staging the tile through LDS and without its 8 workgroup barriers.  Interior tiles of the
// specialised kinds only; everything else takes gemm_epilogue<..., SWP = true>.
// ---------------------------------------------------------------------------------------------
template <int KIND>
MTT_DEV void gemm_epilogue_direct(const GemmP& p, f32x4 (&acc)[8][4], int m0, int n0, int zo, int zi) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int wm = wave >> 2, wn = wave & 3;
  const mtt_gemm_desc& d = p.d;
  const int col0 = n0 + wn * 64 + (lg & 1) * 16 + (lg >> 1) * 8;      // + 32 * pair
  float sh[2][8], cs[2][8];
  {
    const int64_t zcol = (int64_t)zo * d.col_zo + (int64_t)zi * d.col_zi + col0;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        sh[q][j] = d.colshift ? d.colshift[zcol + q * 32 + j] : 0.0f;
        cs[q][j] = d.colscale ? d.colscale[zcol + q * 32 + j] : 1.0f;
      }
  }
  const int64_t zD = (int64_t)zo * d.d_zo + (int64_t)zi * d.d_zi + col0;
  const int64_t zAux = (int64_t)zo * d.aux_zo + (int64_t)zi * d.aux_zi + col0;
  const int64_t zR = (int64_t)zo * d.r_zo + (int64_t)zi * d.r_zi + col0;
  const int mbase = m0 + wm * 128 + li;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    // this half's residual / GELU' input rows, all issued before any store of the half (D and resid may alias)
    float4 ra[4][2], rb[4][2];
    u32x4 za[4][2];
#pragma unroll
    for (int aa = 0; aa < 4; ++aa)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int64_t m = mbase + (half * 4 + aa) * 16;
        if (KIND == 3) {
          if (d.resid) { ra[aa][q] = *(const float4*)(d.resid + (zR + m * d.ldr + q * 32)); rb[aa][q] = *(const float4*)(d.resid + (zR + m * d.ldr + q * 32) + 4); }
          else { ra[aa][q] = rb[aa][q] = make_float4(0.f, 0.f, 0.f, 0.f); }
        }
        if (KIND == 4) za[aa][q] = *(const u32x4*)((const bf16_t*)d.aux_in + (zAux + m * d.ldaux + q * 32));
      }
#pragma unroll
    for (int aa = 0; aa < 4; ++aa) {
      const int a = half * 4 + aa;
      const int64_t m = mbase + a * 16;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          // inline asm, not __builtin_amdgcn_permlane16_swap: hipcc (ROCm 7.2) folds the four builtin calls of this unrolled loop into ONE
          // swap and replicates its result (seen in the .s and in a 10-line reproducer); the s_nop covers the VALU-write -> permlane hazard
          float lo = acc[a][2 * q][r], hi = acc[a][2 * q + 1][r];
          asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));
          v[r] = fmaf(lo, cs[q][r], sh[q][r]);
          v[4 + r] = fmaf(hi, cs[q][4 + r], sh[q][4 + r]);
        }
        if (KIND == 2) {
          if (d.aux_out)
            *(u32x4*)((bf16_t*)d.aux_out + (zAux + m * d.ldaux + q * 32)) = (u32x4){pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = gelu_f(v[j]);
        }
        if (KIND == 4) {
          const u32x4 u = za[aa][q];
          v[0] *= gelu_grad_f(lo_of(u.x)); v[1] *= gelu_grad_f(hi_of(u.x)); v[2] *= gelu_grad_f(lo_of(u.y)); v[3] *= gelu_grad_f(hi_of(u.y));
          v[4] *= gelu_grad_f(lo_of(u.z)); v[5] *= gelu_grad_f(hi_of(u.z)); v[6] *= gelu_grad_f(lo_of(u.w)); v[7] *= gelu_grad_f(hi_of(u.w));
        }
        if (KIND == 3) {
          if (d.rowscale) {
            const uint32_t qq = fdiv((uint32_t)m, p.divDmb), rem = (uint32_t)m - qq * (uint32_t)d.d_mb;
            const float rs = d.rowscale[qq * 2 + (rem >= (uint32_t)d.n_prompt ? 1 : 0)];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] *= rs;
          }
          v[0] += ra[aa][q].x; v[1] += ra[aa][q].y; v[2] += ra[aa][q].z; v[3] += ra[aa][q].w;
          v[4] += rb[aa][q].x; v[5] += rb[aa][q].y; v[6] += rb[aa][q].z; v[7] += rb[aa][q].w;
        }
        if (KIND == 1 || KIND == 3) {
          float* dp = (float*)d.D + (zD + m * d.ldd + q * 32);
          *(float4*)dp = make_float4(v[0], v[1], v[2], v[3]);
          *(float4*)(dp + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else {
          *(u32x4*)((bf16_t*)d.D + (zD + m * d.ldd + q * 32)) = (u32x4){pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
        }
      }
    }
  }
}

template <int TBN, int WAVES_M, int WAVES_N, int MT, int NTL>
MTT_DEV void gemm_epilogue_auto_swapped(const GemmP& p, f32x4 (&acc)[MT][NTL], unsigned char* smem, int m0, int n0, int zo, int zi) {
  static_assert(TBN == 256 && WAVES_M == 2 && WAVES_N == 4 && MT == 8 && NTL == 4, "direct epilogue: 256 x 256 tile, 2 x 4 waves");
  const int kind = fast_epilogue_kind(p.d, m0, n0, 256, 256);
  if (kind == 0) gemm_epilogue_direct<0>(p, acc, m0, n0, zo, zi);
  else if (kind == 1) gemm_epilogue_direct<1>(p, acc, m0, n0, zo, zi);
  else if (kind == 2) gemm_epilogue_direct<2>(p, acc, m0, n0, zo, zi);
  else if (kind == 3) gemm_epilogue_direct<3>(p, acc, m0, n0, zo, zi);
  else if (kind == 4) gemm_epilogue_direct<4>(p, acc, m0, n0, zo, zi);
  else gemm_epilogue<TBN, WAVES_M, WAVES_N, MT, NTL, 0, true>(p, acc, smem, m0, n0, zo, zi);
}

// epilogue dispatch (workgroup-uniform): specialised path for interior tiles of the hot call sites, general path otherwise
template <int TBN, int WAVES_M, int WAVES_N, int MT, int NTL>
MTT_DEV void gemm_epilogue_auto(const GemmP& p, f32x4 (&acc)[MT][NTL], unsigned char* smem, int m0, int n0, int zo, int zi) {
  const int kind = p.d.variant == MTT_GEMM_GENERAL_EPILOGUE ? -1 : fast_epilogue_kind(p.d, m0, n0, WAVES_M * MT * 16, TBN);
  if (kind == 0) gemm_epilogue_fast<0, TBN, WAVES_M, WAVES_N, MT, NTL>(p, acc, smem, m0, n0, zo, zi);
  else if (kind == 1) gemm_epilogue_fast<1, TBN, WAVES_M, WAVES_N, MT, NTL>(p, acc, smem, m0, n0, zo, zi);
  else if (kind == 2) gemm_epilogue_fast<2, TBN, WAVES_M, WAVES_N, MT, NTL>(p, acc, smem, m0, n0, zo, zi);
  else if (kind == 3) gemm_epilogue_fast<3, TBN, WAVES_M, WAVES_N, MT, NTL>(p, acc, smem, m0, n0, zo, zi);
  else if (kind == 4) gemm_epilogue_fast<4, TBN, WAVES_M, WAVES_N, MT, NTL>(p, acc, smem, m0, n0, zo, zi);
  else gemm_epilogue<TBN, WAVES_M, WAVES_N, MT, NTL>(p, acc, smem, m0, n0, zo, zi);
}

// ---------------------------------------------------------------------------------------------
// MODE: 0 = bf16 MFMA, A and B bf16;  1 = bf16 MFMA, A f32 (converted while staging), B bf16;  2 = X3 (both f32)
template <int AOP, int BOP, int MODE>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmP p) {
  constexpr bool X3 = MODE == 2, AF32 = MODE >= 1, BF32 = MODE == 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NPL = X3 ? 2 : 1;
  constexpr int STAGE = TILE_BYTES * 2 * NPL;   // A planes then B planes

  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  int tile_m, tile_n;
  grouped_tile(wg, p.tiles_m, p.tiles_n, p.group_m, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int z = blockIdx.z;
  const int zo = z / p.d.batch_inner, zi = z - zo * p.d.batch_inner;

  const int esA = p.d.a_dtype == MTT_F32 ? 4 : 2, esB = p.d.b_dtype == MTT_F32 ? 4 : 2;
  const void* Abase = (const unsigned char*)p.d.A + ((int64_t)zo * p.d.a_zo + (int64_t)zi * p.d.a_zi) * esA;
  const void* Bbase = (const unsigned char*)p.d.B + ((int64_t)zo * p.d.b_zo + (int64_t)zi * p.d.b_zi) * esB;

  typename StagerSel<AOP, X3, AF32>::type sa;
  typename StagerSel<BOP, X3, BF32>::type sb;
  stager_init_a<AOP, X3>(sa, p, Abase, m0);
  stager_init_b<BOP, X3>(sb, p, Bbase, n0);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;

  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = (p.d.K + BK - 1) / BK;
  sa.load(p, 0); sb.load(p, 0);
  {
    unsigned char* st = smem;
    sa.store(st, st + TILE_BYTES);
    sb.store(st + TILE_BYTES * NPL, st + TILE_BYTES * NPL + TILE_BYTES);
  }
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    if (more) { sa.load(p, (kt + 1) * BK); sb.load(p, (kt + 1) * BK); }

    const unsigned char* st = smem + (kt & 1) * STAGE;
    const unsigned char* Ah = st;
    const unsigned char* Al = st + TILE_BYTES;
    const unsigned char* Bh = st + TILE_BYTES * NPL;
    const unsigned char* Bl = Bh + TILE_BYTES;
    if constexpr (!X3) {
      // all 16 fragment reads of this K step are issued before the 32 MFMAs (the compiler then places
      // counted lgkmcnt waits, so LDS latency overlaps the first MFMAs instead of stalling every group)
      u32x4 fa[2][4], fb[2][4];
#pragma unroll
      for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          fa[kh][t] = *(const u32x4*)(Ah + lds_off(wm * 64 + t * 16 + li, kh * 4 + lg));
          fb[kh][t] = *(const u32x4*)(Bh + lds_off(wn * 64 + t * 16 + li, kh * 4 + lg));
        }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) acc[a][b] = mfma16(fa[kh][a], fb[kh][b], acc[a][b]);
    } else {
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
        u32x4 ah[4], al[4], bh[4], bl[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int ra = wm * 64 + t * 16 + li, rbn = wn * 64 + t * 16 + li;
          ah[t] = *(const u32x4*)(Ah + lds_off(ra, kh * 4 + lg));
          bh[t] = *(const u32x4*)(Bh + lds_off(rbn, kh * 4 + lg));
          al[t] = *(const u32x4*)(Al + lds_off(ra, kh * 4 + lg));
          bl[t] = *(const u32x4*)(Bl + lds_off(rbn, kh * 4 + lg));
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            acc[a][b] = mfma16(al[a], bh[b], acc[a][b]);
            acc[a][b] = mfma16(ah[a], bl[b], acc[a][b]);
            acc[a][b] = mfma16(ah[a], bh[b], acc[a][b]);
          }
      }
    }

    if (more) {
      unsigned char* sn = smem + ((kt + 1) & 1) * STAGE;
      sa.store(sn, sn + TILE_BYTES);
      sb.store(sn + TILE_BYTES * NPL, sn + TILE_BYTES * NPL + TILE_BYTES);
    }
    __syncthreads();
  }

  gemm_epilogue_auto<BN, 2, 2, 4, 4>(p, acc, smem, m0, n0, zo, zi);
}


// ---------------------------------------------------------------------------------------------
// Fast path: both operands bf16 and reduction-contiguous, K % 64 == 0.  Tiles go HBM -> LDS directly
// (global_load_lds_dwordx4, 1 KiB per wave-instruction; the XOR swizzle is applied to the per-lane SOURCE chunk
// so the LDS image stays lane-linear), through a ring of FSTAGES stages with counted vmcnt waits and raw barriers:
// FSTAGES-2 tiles stay in flight across every barrier, one barrier per K step, no staging registers.
// ---------------------------------------------------------------------------------------------
constexpr int FSTAGES = 4;

MTT_DEV void glds16(const bf16_t* src, unsigned char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__global__ __launch_bounds__(256, 1) void gemm_fast_kernel(const GemmP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int STAGE = TILE_BYTES * 2;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  int tile_m, tile_n;
  grouped_tile(wg, p.tiles_m, p.tiles_n, p.group_m, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int z = blockIdx.z;
  const int zo = z / p.d.batch_inner, zi = z - zo * p.d.batch_inner;
  const bf16_t* Abase = (const bf16_t*)p.d.A + ((int64_t)zo * p.d.a_zo + (int64_t)zi * p.d.a_zi);
  const bf16_t* Bbase = (const bf16_t*)p.d.B + ((int64_t)zo * p.d.b_zo + (int64_t)zi * p.d.b_zi);

  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;

  // per-lane source pointers: wave w streams rows [32w, 32w+32) of each tile as 4 x (8 rows x 128 B)
  const bf16_t* pa[4];
  const bf16_t* pb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ lds_swz(row);
    int ra = m0 + row; if (ra > p.d.M - 1) ra = p.d.M - 1;      // ragged edge: re-read the last valid row (results unused)
    int rb = n0 + row; if (rb > p.d.N - 1) rb = p.d.N - 1;
    pa[i] = Abase + row_off((uint32_t)ra, p.d.a_mb, p.d.a_bs, p.d.lda, p.divAmb) + c * 8;
    pb[i] = Bbase + (int64_t)rb * p.d.ldb + c * 8;
  }
  auto issue = [&](int stage, int kt) {
    unsigned char* sA = smem + stage * STAGE + wave * 4096;
    unsigned char* sB = sA + TILE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      glds16(pa[i] + kt * BK, sA + i * 1024);
      glds16(pb[i] + kt * BK, sB + i * 1024);
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = p.d.K / BK;
#pragma unroll
  for (int s = 0; s < FSTAGES - 1; ++s)
    if (s < nk) issue(s, s);

  for (int kt = 0; kt < nk; ++kt) {
    // tile kt has landed once at most min(FSTAGES-2, tiles issued after it) x 8 of this wave's loads are outstanding
    const int after = nk - 1 - kt;
    if (after >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (after == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();          // every wave's part of tile kt is in LDS; everyone is done reading stage (kt-1)
    if (kt + FSTAGES - 1 < nk) issue((kt + FSTAGES - 1) % FSTAGES, kt + FSTAGES - 1);

    const unsigned char* Ah = smem + (kt % FSTAGES) * STAGE;
    const unsigned char* Bh = Ah + TILE_BYTES;
    u32x4 fa[2][4], fb[2][4];
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        fa[kh][t] = *(const u32x4*)(Ah + lds_off(wm * 64 + t * 16 + li, kh * 4 + lg));
        fb[kh][t] = *(const u32x4*)(Bh + lds_off(wn * 64 + t * 16 + li, kh * 4 + lg));
      }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = mfma16(fa[kh][a], fb[kh][b], acc[a][b]);
  }
  __syncthreads();                          // all waves done with the last stage before LDS is reused by the epilogue
  gemm_epilogue_auto<BN, 2, 2, 4, 4>(p, acc, smem, m0, n0, zo, zi);
}

int launch_fast(const GemmP& p, hipStream_t stream) {
  constexpr int smem = TILE_BYTES * 2 * FSTAGES;
  static std::atomic<unsigned long long> done{0};
  if (int e = mtt_ensure_dyn_lds((const void*)gemm_fast_kernel, smem, done)) return e;
  dim3 grid(p.tiles_m * p.tiles_n, 1, p.d.batch);
  hipLaunchKernelGGL(gemm_fast_kernel, grid, dim3(256), smem, stream, p);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// 256 x 256 x 64 tile, 8 waves (2 x 4, each 128 x 64 = 8 x 4 MFMA tiles, 128 fp32 accumulators per lane), operands
// streamed HBM -> LDS by global_load_lds into 2 stages of 64 KiB.  Twice the arithmetic intensity of the 128 x 128
// tile (128 FLOP per staged byte): the 128-tile kernels need ~64 B/clk/CU from L2 to keep the MFMA pipe busy,
// which L2 cannot deliver; this one needs 32.
// ---------------------------------------------------------------------------------------------
constexpr int BM2 = 256, BN2 = 256, TILE2 = BM2 * BK * 2;     // 32 KiB per operand tile

__global__ __launch_bounds__(512, 1) void gemm_fast256_kernel(const GemmP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int STAGE = TILE2 * 2;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int tiles_n = (p.d.N + BN2 - 1) / BN2, tiles_m = (p.d.M + BM2 - 1) / BM2;
  int tile_m, tile_n;
  grouped_tile(wg, tiles_m, tiles_n, p.group_m, tile_m, tile_n);
  const int m0 = tile_m * BM2, n0 = tile_n * BN2;
  const int z = blockIdx.z;
  const int zo = z / p.d.batch_inner, zi = z - zo * p.d.batch_inner;
  const bf16_t* Abase = (const bf16_t*)p.d.A + ((int64_t)zo * p.d.a_zo + (int64_t)zi * p.d.a_zi);
  const bf16_t* Bbase = (const bf16_t*)p.d.B + ((int64_t)zo * p.d.b_zo + (int64_t)zi * p.d.b_zi);

  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int wm = wave >> 2, wn = wave & 3;

  const bf16_t* pa[4];
  const bf16_t* pb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 8 + (lane >> 3);          // wave w streams rows [32w, 32w + 32)
    const int c = (lane & 7) ^ lds_swz(row);
    int ra = m0 + row; if (ra > p.d.M - 1) ra = p.d.M - 1;
    int rb = n0 + row; if (rb > p.d.N - 1) rb = p.d.N - 1;
    pa[i] = Abase + row_off((uint32_t)ra, p.d.a_mb, p.d.a_bs, p.d.lda, p.divAmb) + c * 8;
    pb[i] = Bbase + (int64_t)rb * p.d.ldb + c * 8;
  }
  auto issue = [&](int stage, int kt) {
    unsigned char* sA = smem + stage * STAGE + wave * 4096;
    unsigned char* sB = sA + TILE2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      glds16(pa[i] + kt * BK, sA + i * 1024);
      glds16(pb[i] + kt * BK, sB + i * 1024);
    }
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = p.d.K / BK;
  issue(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's part of tile kt has landed
    __builtin_amdgcn_s_barrier();                           // ... everyone's has, and everyone finished reading stage (kt+1)&1
    if (kt + 1 < nk) issue((kt + 1) & 1, kt + 1);
    const unsigned char* Ah = smem + (kt & 1) * STAGE;
    const unsigned char* Bh = Ah + TILE2;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      u32x4 fa[8], fb[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) fb[t] = *(const u32x4*)(Bh + lds_off(wn * 64 + t * 16 + li, kh * 4 + lg));
#pragma unroll
      for (int t = 0; t < 8; ++t) fa[t] = *(const u32x4*)(Ah + lds_off(wm * 128 + t * 16 + li, kh * 4 + lg));
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = mfma16(fa[a], fb[b], acc[a][b]);
      __builtin_amdgcn_s_setprio(0);
    }
  }
  __syncthreads();
  gemm_epilogue_auto<BN2, 2, 4, 8, 4>(p, acc, smem, m0, n0, zo, zi);
}

int launch_fast256(const GemmP& p, hipStream_t stream) {
  constexpr int smem = TILE2 * 2 * 2;
  static std::atomic<unsigned long long> done{0};
  if (int e = mtt_ensure_dyn_lds((const void*)gemm_fast256_kernel, smem, done)) return e;
  const int tm = (p.d.M + BM2 - 1) / BM2, tn = (p.d.N + BN2 - 1) / BN2;
  dim3 grid(tm * tn, 1, p.d.batch);
  hipLaunchKernelGGL(gemm_fast256_kernel, grid, dim3(512), smem, stream, p);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// gemm_dma_kernel<BN, CONV>: the round-2 main kernel.  256 x BN x 64 tile (BN = 256 or 128), 8 waves, operands streamed
// HBM -> LDS by global_load_lds into 2 stages, like gemm_fast256_kernel, but with a PHASED, STAGGERED main loop:
//
//   * a K step is four phases  R0 | C0 | R1 | C1  (R = ds_read the fragments of one 32-deep half, C = its MFMAs), each closed by
//     a workgroup barrier;
//   * waves 4-7 run ONE phase behind waves 0-3 (one extra barrier before the loop, one after it for the other half).  Wave w and
//     wave w+4 share a SIMD, so in every slot one of the SIMD's two waves reads LDS while the other issues MFMAs: fragment reads
//     no longer alternate with an idle matrix pipe (the lock-step 2-stage loop measured ~56 % MFMA issue at K = 8192);
//   * the next K tile is issued (LDS-DMA) in R0 and waited for (vmcnt(0)) at the end of R1, i.e. it has R0 + C0 + R1 to land;
//     the barrier closing R1 of the later half precedes the first read of that tile by the earlier half.
//   Hazards (slot numbering: waves 0-3 run phase p of K step kt in slot 4 kt + p, waves 4-7 in slot 4 kt + p + 1):
//     RAW  tile kt+1 is first read in slot 4 kt + 4; its last writer waits vmcnt(0) in slot 4 kt + 3, a barrier separates them.
//     WAR  stage (kt+1)&1 is overwritten from slot 4 kt on; its last readers (tile kt-1, R1) ran in slots 4 kt - 2 / 4 kt - 1 and
//          waited lgkmcnt(0) before their closing barrier.
// CONV: A is the implicit im2col of a 3x3 (dilated) convolution; every lane computes the source address of its 16-byte channel
// chunk per K step (tap, channel tracked incrementally; halo / K tail chunks read a zero page) — address VALU work runs in the
// R phases, under the other half's MFMAs.  K only needs to be a multiple of 8 (chunks past K read the zero page).
// ---------------------------------------------------------------------------------------------
// SCHED selects where the LDS-DMA pieces of the next K tile are issued (an LDS-DMA instruction costs its wave ~60-100 cycles of issue;
// eight of them in one burst made R0 twice as long as a C phase):
//   0  all 8 pieces in R0 (first version: profiles/r02_gemm_bench_b*.log — no gain over the lock-step kernel)
//   1  balanced: waves 0-3 issue the A pieces in R0 and the B pieces in R1; waves 4-7 (one slot behind) issue the A pieces of tile
//      kt+2 between the MFMAs of C1(kt) and the B pieces of tile kt+1... see the schedule table in the loop body.
template <int BN_, bool CONV, int SCHED>
__global__ __launch_bounds__(512, 1) void gemm_dma_kernel(const GemmP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int WAVES_N = BN_ / 64, WAVES_M = 8 / WAVES_N, MT = 256 / WAVES_M / 16, NT = 4;
  constexpr int TILE_A = BM2 * BK * 2, TILE_B = BN_ * BK * 2, STAGE = TILE_A + TILE_B;
  constexpr int B_GLDS = BN_ / 64;                 // 1 KiB pieces of the B tile per wave (8 rows x 128 B each)
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int tiles_n = (p.d.N + BN_ - 1) / BN_, tiles_m = (p.d.M + BM2 - 1) / BM2;
  int tile_m, tile_n;
  grouped_tile(wg, tiles_m, tiles_n, p.group_m, tile_m, tile_n);
  const int m0 = tile_m * BM2, n0 = tile_n * BN_;
  const int z = blockIdx.z;
  const int zo = z / p.d.batch_inner, zi = z - zo * p.d.batch_inner;
  const bf16_t* Abase = (const bf16_t*)p.d.A + ((int64_t)zo * p.d.a_zo + (int64_t)zi * p.d.a_zi);
  const bf16_t* Bbase = (const bf16_t*)p.d.B + ((int64_t)zo * p.d.b_zo + (int64_t)zi * p.d.b_zi);

  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int late = wave >> 2;                      // 1: this wave runs one phase behind (shares its SIMD with wave - 4)
  const int K = p.d.K;

  uint64_t zpage = (uint64_t)(uintptr_t)g_zero_page;
  asm volatile("" : "+s"(zpage));

  // ---- per-lane source addressing ------------------------------------------------------------------------------------
  // A: wave w streams rows [32 w, 32 w + 32) as 4 pieces of 8 rows x 128 B; lane -> (row, swizzled 16-byte chunk)
  int64_t aoff[4];                                 // element offset of (row, chunk 0) [plain: + chunk]   (CONV: of the output pixel's row)
  int ack[4];                                      // this lane's k offset inside a K step (chunk * 8)
  unsigned tapmask[4]; int tap[4], ci[4];          // CONV only
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ lds_swz(row);
    ack[i] = c * 8;
    int ra = m0 + row; if (ra > p.d.M - 1) ra = p.d.M - 1;      // ragged edge: re-read the last valid row (results unused)
    if constexpr (CONV) {
      const uint32_t t = fdiv((uint32_t)ra, p.divW);
      const int px = ra - (int)t * p.d.conv.W;
      const uint32_t bb = fdiv(t, p.divH);
      const int py = (int)t - (int)bb * p.d.conv.H;
      unsigned m = 0;
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) {
        int ty = tp / 3, tx = tp % 3;
        if (p.d.conv.flip) { ty = 2 - ty; tx = 2 - tx; }
        const int yy = py + (ty - 1) * p.d.conv.dil, xx = px + (tx - 1) * p.d.conv.dil;
        if (yy >= 0 && yy < p.d.conv.H && xx >= 0 && xx < p.d.conv.W) m |= 1u << tp;
      }
      tapmask[i] = m;
      aoff[i] = (int64_t)ra * p.d.lda;
      tap[i] = 0; ci[i] = ack[i];
      while (ci[i] >= p.d.conv.Cp) { ci[i] -= p.d.conv.Cp; ++tap[i]; }
    } else {
      aoff[i] = row_off((uint32_t)ra, p.d.a_mb, p.d.a_bs, p.d.lda, p.divAmb) + ack[i];
      tapmask[i] = 0; tap[i] = 0; ci[i] = 0;
    }
  }
  // B: BN_ rows; wave w streams rows [BN_/8 * w, ...) as B_GLDS pieces of 8 rows
  int64_t boff[B_GLDS]; int bck[B_GLDS];
#pragma unroll
  for (int i = 0; i < B_GLDS; ++i) {
    const int row = (wave * B_GLDS + i) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ lds_swz(row);
    bck[i] = c * 8;
    int rb = n0 + row; if (rb > p.d.N - 1) rb = p.d.N - 1;
    boff[i] = (int64_t)rb * p.d.ldb + bck[i];
  }

  // SCHED 7 = SCHED 0 for calls with K % 64 == 0 and operand spans < 2 GiB (host-checked): the source address of a piece is a
  // wave-uniform base (advanced by the K offset on the scalar unit) + a constant 32-bit per-lane byte offset, and there is no K tail,
  // so a piece costs one 64-bit add instead of two adds, a compare and two selects (58 -> 32 VALU instructions per K step; measured
  // on the persistent kernel's identical loop: 1 469 vs 1 395 TFLOP/s at 8192^3 without epilogue, profiles/r02_gemm_ablate_o_*.log)
  constexpr bool FASTADDR = (SCHED == 7 || SCHED == 8) && !CONV;
  constexpr bool SWAPPED = SCHED == 8 && BN_ == 256 && !CONV;      // MFMAs with swapped operands + the direct-store epilogue
  uint32_t aoff32[4], boff32[B_GLDS];
  if constexpr (FASTADDR) {
#pragma unroll
    for (int i = 0; i < 4; ++i) aoff32[i] = (uint32_t)aoff[i] * 2u;
#pragma unroll
    for (int i = 0; i < B_GLDS; ++i) boff32[i] = (uint32_t)boff[i] * 2u;
  }
  auto issueA = [&](int stage, int kt) {
    unsigned char* sA = smem + stage * STAGE + wave * 4096;
    const int k0 = kt * BK;
    if constexpr (FASTADDR) {
      const unsigned char* Ak = (const unsigned char*)Abase + (size_t)kt * (BK * 2);
#pragma unroll
      for (int i = 0; i < 4; ++i) glds16((const bf16_t*)(Ak + aoff32[i]), sA + i * 1024);
      return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uint64_t src;
      bool ok = k0 + ack[i] < K;
      if constexpr (CONV) {
        int ty = (tap[i] * 11) >> 5, tx = tap[i] - 3 * ty;             // tap / 3, tap % 3 for tap < 9 (garbage beyond K: masked by ok)
        if (p.d.conv.flip) { ty = 2 - ty; tx = 2 - tx; }
        const int shift = ((ty - 1) * p.d.conv.W + (tx - 1)) * p.d.conv.dil;
        ok = ok && ((tapmask[i] >> tap[i]) & 1u);
        src = (uint64_t)(uintptr_t)(Abase + (aoff[i] + (int64_t)shift * p.d.lda + ci[i]));
        ci[i] += BK;
        while (ci[i] >= p.d.conv.Cp) { ci[i] -= p.d.conv.Cp; ++tap[i]; }
      } else {
        src = (uint64_t)(uintptr_t)(Abase + (aoff[i] + k0));
      }
      glds16((const bf16_t*)(uintptr_t)(ok ? src : zpage), sA + i * 1024);
    }
  };
  auto issueB = [&](int stage, int kt) {
    unsigned char* sB = smem + stage * STAGE + TILE_A + wave * (B_GLDS * 1024);
    const int k0 = kt * BK;
    if constexpr (FASTADDR) {
      const unsigned char* Bk = (const unsigned char*)Bbase + (size_t)kt * (BK * 2);
#pragma unroll
      for (int i = 0; i < B_GLDS; ++i) glds16((const bf16_t*)(Bk + boff32[i]), sB + i * 1024);
      return;
    }
#pragma unroll
    for (int i = 0; i < B_GLDS; ++i) {
      const bool ok = k0 + bck[i] < K;
      const uint64_t src = (uint64_t)(uintptr_t)(Bbase + (boff[i] + k0));
      glds16((const bf16_t*)(uintptr_t)(ok ? src : zpage), sB + i * 1024);
    }
  };
  auto issue = [&](int stage, int kt) { issueA(stage, kt); issueB(stage, kt); };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // SCHED 2 / 3 are measurement-only ablations (cdna_hip_programming.md §5.4 rule 17): 2 = main loop without the epilogue (accumulators
  // kept alive, nothing stored), 3 = prologue + epilogue without the K loop (one K tile).  Never dispatched by the policy.
  const int nk = SCHED == 3 ? 1 : (K + BK - 1) / BK;
  if (SCHED == 4 && blockIdx.x < 256 && blockIdx.z == 0) {
    // experiment: phase skew.  Every tile of a round finishes at the same time, so all 256 CUs store their tiles at once (a
    // 33-134 MB burst at the fabric's bandwidth limit: profiles/r02_gemm_ablate_d_tile_time.log) while HBM idles during the K loops.
    // Delaying the first-round workgroups by 0..7 x ~1.5 us spreads the later rounds' epilogues over time.
    const int steps = blockIdx.x & 7;
    for (int i = 0; i < steps; ++i) __builtin_amdgcn_s_sleep(47);
  }
  issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                    // tile 0 is in LDS
  if (SCHED == 1 && late && nk > 1) issueA(1, 1);  // what C1(-1) would have issued
  if (late) __builtin_amdgcn_s_barrier();          // stagger: waves 4-7 start one slot later
  __builtin_amdgcn_sched_barrier(0);

  // Schedule of the LDS-DMA of tile kt+1 (SCHED 1); slot = 4 kt + phase (+1 for the late half); deadline = the barrier closing slot 4 kt + 3:
  //   waves 0-3:  A pieces in R0(kt) [slot 4kt], B pieces in R1(kt) [slot 4kt+2], vmcnt(0) at the end of C1(kt) [slot 4kt+3]
  //   waves 4-7:  A pieces in C1(kt-1) [slot 4kt] between its MFMAs, B pieces in R0(kt) [slot 4kt+1], vmcnt(0) at the end of R1(kt) [4kt+3]
  //   (the stage being written held tile kt-1, whose last reads — R1(kt-1) of the late half — ended in slot 4kt-1.)
  // The loop body is instantiated once per half (LATE is a compile-time constant inside it: no per-phase branching, and the
  // register allocator sees one straight-line schedule per half).
  auto main_loop = [&](auto late_tag) {
    constexpr bool LATE = decltype(late_tag)::value;
    for (int kt = 0; kt < nk; ++kt) {
      const unsigned char* Ah = smem + (kt & 1) * STAGE;
      const unsigned char* Bh = Ah + TILE_A;
      const bool more = kt + 1 < nk;
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
        // ---- R phase: fragments of this 32-deep half (+ this wave's share of the next tile's LDS-DMA) ----
        if (SCHED != 1) {
          if (kh == 0 && more) issue((kt + 1) & 1, kt + 1);
        } else if (more) {
          if (kh == 0) { if (!LATE) issueA((kt + 1) & 1, kt + 1); else issueB((kt + 1) & 1, kt + 1); }
          else if (!LATE) issueB((kt + 1) & 1, kt + 1);
        }
        u32x4 fa[MT], fb[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) fb[t] = *(const u32x4*)(Bh + lds_off(wn * 64 + t * 16 + li, kh * 4 + lg));
#pragma unroll
        for (int t = 0; t < MT; ++t) fa[t] = *(const u32x4*)(Ah + lds_off(wm * (MT * 16) + t * 16 + li, kh * 4 + lg));
        if (kh == 1 && (SCHED != 1 || LATE)) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // next tile (own part) landed
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- C phase ----
        __builtin_amdgcn_s_setprio(1);
        if (SCHED == 1 && kh == 1 && LATE && !CONV) {
          // late half: the A pieces of tile kt+2 between the MFMAs (stage kt&1: its last reads, R1(kt), ended before this slot)
          unsigned char* sA = smem + (kt & 1) * STAGE + wave * 4096;
          const int k0 = (kt + 2) * BK;
          const bool more2 = kt + 2 < nk;
#pragma unroll
          for (int a = 0; a < MT; ++a) {
#pragma unroll
            for (int b = 0; b < NT; ++b) acc[a][b] = mfma16(fa[a], fb[b], acc[a][b]);
            if ((a & 1) == 1 && more2) {
              const int i = a >> 1;
              const bool ok = k0 + ack[i] < K;
              const uint64_t src = (uint64_t)(uintptr_t)(Abase + (aoff[i] + k0));
              glds16((const bf16_t*)(uintptr_t)(ok ? src : zpage), sA + i * 1024);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {
#pragma unroll
          for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int b = 0; b < NT; ++b) acc[a][b] = SWAPPED ? mfma16(fb[b], fa[a], acc[a][b]) : mfma16(fa[a], fb[b], acc[a][b]);
          if (SCHED == 1 && kh == 1 && LATE && CONV && kt + 2 < nk) issueA(kt & 1, kt + 2);
        }
        __builtin_amdgcn_s_setprio(0);
        if (SCHED == 1 && kh == 1 && !LATE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // early half: tile kt+1 landed
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  if (late) main_loop(std::true_type{}); else main_loop(std::false_type{});
  if (!late) __builtin_amdgcn_s_barrier();         // waves 0-3 wait one slot for the late half
  __syncthreads();                                 // everyone is past its last LDS read: the epilogue may reuse the stages
  if (SCHED == 2) {
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int b = 0; b < NT; ++b) asm volatile("" :: "v"(acc[a][b]));
    return;
  }
  if constexpr (SCHED == 5 || SCHED == 6 || SCHED == 3)
    gemm_epilogue<BN_, WAVES_M, WAVES_N, MT, NT, (SCHED == 5 ? 1 : (SCHED == 6 ? 2 : 0))>(p, acc, smem, m0, n0, zo, zi);
  else if constexpr (SWAPPED) {
    if (p.d.variant == MTT_GEMM_GENERAL_EPILOGUE) gemm_epilogue<BN_, WAVES_M, WAVES_N, MT, NT, 0, true>(p, acc, smem, m0, n0, zo, zi);
    else gemm_epilogue_auto_swapped<BN_, WAVES_M, WAVES_N, MT, NT>(p, acc, smem, m0, n0, zo, zi);
  } else
    gemm_epilogue_auto<BN_, WAVES_M, WAVES_N, MT, NT>(p, acc, smem, m0, n0, zo, zi);
}

template <int BN_, bool CONV, int SCHED>
int launch_dma(const GemmP& p, hipStream_t stream) {
  constexpr int smem = (BM2 * BK * 2 + BN_ * BK * 2) * 2;
  static std::atomic<unsigned long long> done{0};
  if (int e = mtt_ensure_dyn_lds((const void*)gemm_dma_kernel<BN_, CONV, SCHED>, smem, done)) return e;
  const int tm = (p.d.M + BM2 - 1) / BM2, tn = (p.d.N + BN_ - 1) / BN_;
  dim3 grid(tm * tn, 1, p.d.batch);
  hipLaunchKernelGGL((gemm_dma_kernel<BN_, CONV, SCHED>), grid, dim3(512), smem, stream, p);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// gemm_pdma_kernel<KIND, V>: PERSISTENT form of gemm_dma_kernel<256, false, 0> for the hot encoder GEMMs (plain bf16 operands,
// N % 256 == 0, K % 64 == 0, K >= 128, one of the specialised epilogue kinds).  Measured on the one-tile-per-workgroup kernel
// (profiles/r02_gemm_ablate_n_*.log): of a 35.5 us K = 1024 tile 26 us are K steps; ~2 us are the prologue (workgroup launch,
// address setup, the first LDS-DMA round trip with nothing to overlap it) and ~7.7 us the epilogue, whose accumulators go through
// the operand stages (so nothing can be prefetched under it) with 8 workgroup barriers.  Here:
//   * a workgroup walks tiles f = blockIdx.x, + gridDim.x, ... (grid = one workgroup per CU; the XCD-aware grouped tile order is
//     applied to the flat tile index, so a workgroup keeps its XCD's share of the order);
//   * the K loop is CONTINUOUS across output tiles: the R0 phase of a tile's last K step (the "seam" step, a separate instantiation
//     so that the hot loop carries no tile bookkeeping) streams the NEXT tile's first K tile into the free stage, exactly where a
//     longer K loop would have issued tile kt + 1 (same hazards, see gemm_dma_kernel), and loads the tile's bias / column scale;
//   * the epilogue runs out of a 4 KiB PER-WAVE scratch behind the two stages (160 KiB of LDS in all): each wave moves its own
//     128 x 64 accumulator block 16 rows at a time through LDS (ds_write_b32 in the MFMA layout, ds_read_b128 as rows: conflict-free
//     in both directions with a plain 256-byte row pitch) and stores full 128-byte (bf16) / 256-byte (fp32) row segments.  No
//     workgroup barrier, and the stages stay untouched;
//   * V = 1 (deferred stores): vmcnt counts stores as well as LDS-DMA, so a K step's "my pieces landed" wait would also drain the
//     epilogue's stores.  Therefore the next tile's SECOND K tile is issued right after the last stage read (before the epilogue),
//     the epilogue keeps its results in registers (they take the place of the accumulators they came from), waits once for
//     everything outstanding and only then issues all its stores; the next tile's first K step neither issues nor waits, so the
//     stores have ~1.7 K steps of MFMA work to drain under.  V = 0 stores block by block and keeps the plain one-tile-ahead prefetch.
//   * residual / GELU' input rows of block a + 1 are loaded before block a is finished (D and resid alias: the compiler may not move
//     loads over stores itself).
// The early half waits one slot at the end of a tile (as the one-tile kernel does) so that both halves run their epilogues together
// and every wave is past its last stage read; the late half re-staggers at the start of the next tile.  M may be ragged (rows are
// predicated).  Source addresses are a wave-uniform base + a 32-bit per-lane byte offset (host-checked span < 2 GiB).
// ---------------------------------------------------------------------------------------------
// ABL (measurement only, KIND 0): 1 = no epilogue (accumulators kept alive), 2 = everything but the global stores, 3 = no bias loads
template <int KIND, int V, int ABL>
struct WaveEpilogue {
  static constexpr bool F32OUT = KIND == 1 || KIND == 3;
  const GemmP& p;
  float sh[4], cs[4];
  float4 of[V && F32OUT ? 8 : 1][4];               // deferred outputs (V = 1)
  u32x2 ob[V && !F32OUT ? 8 : 1][4], oz[V && KIND == 2 ? 8 : 1][4];

  MTT_DEV explicit WaveEpilogue(const GemmP& p_) : p(p_) {}

  // bias / column scale of this lane's 4 output columns (issued in the seam step, waited for by that step's own vmcnt(0))
  MTT_DEV void load_cols(int ncol, int zo, int zi) {
    const mtt_gemm_desc& d = p.d;
    const int64_t zcol = (int64_t)zo * d.col_zo + (int64_t)zi * d.col_zi + ncol;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      sh[j] = (d.colshift && ABL != 3) ? d.colshift[zcol + j] : 0.0f;
      cs[j] = (d.colscale && ABL != 3) ? d.colscale[zcol + j] : 1.0f;
    }
  }

  // acc -> (LDS transpose) -> rows; stores immediately (V = 0) or into of / ob / oz (V = 1)
  MTT_DEV void compute(f32x4 (&acc)[8][4], float* scr, int mw, int ncol, int zo, int zi) {
    if (ABL == 1) {
#pragma unroll
      for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) asm volatile("" :: "v"(acc[a][b]));
      return;
    }
    const int lane = threadIdx.x & 63;
    const int li = lane & 15, lg = lane >> 4;
    const mtt_gemm_desc& d = p.d;
    const int64_t zAux = (int64_t)zo * d.aux_zo + (int64_t)zi * d.aux_zi + ncol;
    const int64_t zR = (int64_t)zo * d.r_zo + (int64_t)zi * d.r_zi + ncol;
    float* const wr = scr + lg * 256 + li;           // acc[a][b][r] -> row lg*4 + r, column b*16 + li
    const float* const rd = scr + lg * 64 + li * 4;  // rows lg + 4 i, columns 4 li .. 4 li + 3
    const int mlast = d.M - 1;
    const bool has_res = KIND == 3 && d.resid != nullptr;

    float4 rn[4]; u32x2 zn[4];                       // residual / GELU' input rows of the block being prefetched
    auto prefetch = [&](int a) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int m = mw + a * 16 + lg + 4 * i; if (m > mlast) m = mlast;
        if (KIND == 3) rn[i] = has_res ? *(const float4*)(d.resid + (zR + (int64_t)m * d.ldr)) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (KIND == 4) zn[i] = *(const u32x2*)((const bf16_t*)d.aux_in + (zAux + (int64_t)m * d.ldaux));
      }
    };
    if (KIND == 3 || KIND == 4) prefetch(0);
#pragma unroll
    for (int a = 0; a < 8; ++a) {
      float4 rc[4]; u32x2 zc[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { if (KIND == 3) rc[i] = rn[i]; if (KIND == 4) zc[i] = zn[i]; }
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) wr[r * 64 + b * 16] = acc[a][b][r];
      float4 v4[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) v4[i] = *(const float4*)(rd + i * 256);
      if ((KIND == 3 || KIND == 4) && a + 1 < 8) prefetch(a + 1);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = mw + a * 16 + lg + 4 * i;
        float v[4] = {fmaf(v4[i].x, cs[0], sh[0]), fmaf(v4[i].y, cs[1], sh[1]), fmaf(v4[i].z, cs[2], sh[2]), fmaf(v4[i].w, cs[3], sh[3])};
        if (KIND == 2) {
          if (V) oz[V ? a : 0][i] = (u32x2){pack2(v[0], v[1]), pack2(v[2], v[3])};
          else if (d.aux_out && m <= mlast) *(u32x2*)((bf16_t*)d.aux_out + (zAux + (int64_t)m * d.ldaux)) = (u32x2){pack2(v[0], v[1]), pack2(v[2], v[3])};
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = gelu_f(v[j]);
        }
        if (KIND == 4) {
          v[0] *= gelu_grad_f(lo_of(zc[i].x)); v[1] *= gelu_grad_f(hi_of(zc[i].x));
          v[2] *= gelu_grad_f(lo_of(zc[i].y)); v[3] *= gelu_grad_f(hi_of(zc[i].y));
        }
        if (KIND == 3) {
          if (d.rowscale) {
            const uint32_t mm = (uint32_t)(m <= mlast ? m : mlast);
            const uint32_t q = fdiv(mm, p.divDmb), rem = mm - q * (uint32_t)d.d_mb;
            const float rs = d.rowscale[q * 2 + (rem >= (uint32_t)d.n_prompt ? 1 : 0)];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] *= rs;
          }
          v[0] += rc[i].x; v[1] += rc[i].y; v[2] += rc[i].z; v[3] += rc[i].w;
        }
        if (V) {
          if (F32OUT) of[V ? a : 0][i] = make_float4(v[0], v[1], v[2], v[3]);
          else ob[V ? a : 0][i] = (u32x2){pack2(v[0], v[1]), pack2(v[2], v[3])};
        } else if (m <= mlast) {
          store_row(m, ncol, zo, zi, make_float4(v[0], v[1], v[2], v[3]), (u32x2){pack2(v[0], v[1]), pack2(v[2], v[3])});
        }
      }
    }
  }

  MTT_DEV void store_row(int m, int ncol, int zo, int zi, float4 vf, u32x2 vb) {
    const mtt_gemm_desc& d = p.d;
    const int64_t off = (int64_t)zo * d.d_zo + (int64_t)zi * d.d_zi + ncol + (int64_t)m * d.ldd;
    if (ABL == 2) asm volatile("" :: "v"(vb), "v"(vf.x));
    else if (F32OUT) *(float4*)((float*)d.D + off) = vf;
    else *(u32x2*)((bf16_t*)d.D + off) = vb;
  }

  // V = 1: all the tile's stores in one burst
  MTT_DEV void store_all(int mw, int ncol, int zo, int zi) {
    if (!V || ABL == 1) return;
    const int lg = (threadIdx.x & 63) >> 4;
    const mtt_gemm_desc& d = p.d;
    const int mlast = d.M - 1;
    const int64_t zAux = (int64_t)zo * d.aux_zo + (int64_t)zi * d.aux_zi + ncol;
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = mw + a * 16 + lg + 4 * i;
        if (m <= mlast) {
          if (KIND == 2 && d.aux_out) *(u32x2*)((bf16_t*)d.aux_out + (zAux + (int64_t)m * d.ldaux)) = oz[V && KIND == 2 ? a : 0][i];
          store_row(m, ncol, zo, zi, of[V && F32OUT ? a : 0][i], ob[V && !F32OUT ? a : 0][i]);
        }
      }
  }
};

template <int KIND, int V, int ABL = 0>
__global__ __launch_bounds__(512, 1) void gemm_pdma_kernel(const GemmP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int MT = 8, NT = 4;
  constexpr int TILE_A = BM2 * BK * 2, STAGE = 2 * TILE_A;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int wm = wave >> 2, wn = wave & 3;
  const int late = wave >> 2;                      // waves 4-7 run one phase behind waves 0-3 (they share SIMDs pairwise)
  const int nk = p.d.K / BK;                       // >= 2 (host-checked)
  const int tiles_n = p.d.N / 256, tiles_m = (p.d.M + 255) / 256;
  const int per_z = tiles_m * tiles_n, total = per_z * p.d.batch;
  float* const scr = (float*)(smem + 2 * STAGE) + wave * 1024;

  // per-lane constants of the staging pattern (row within the tile -> swizzled 16-byte chunk)
  int ack[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 8 + (lane >> 3);
    ack[i] = (((lane & 7) ^ lds_swz(row)) * 8);
  }
  // state of the tile whose operands are being STREAMED (runs ahead of the tile being accumulated at a tile seam).  Source addresses
  // are a wave-uniform base (advanced by the K offset) + a 32-bit per-lane byte offset: no per-piece 64-bit offset arithmetic, and
  // K % 64 == 0 means no K-tail selects either.
  int m0 = 0, n0 = 0, zo = 0, zi = 0;
  const unsigned char* Abase = nullptr; const unsigned char* Bbase = nullptr;
  uint32_t aoff[4], boff[4];
  auto setup = [&](int f) {
    int z = 0, t = f;
    if (p.d.batch > 1) { z = f / per_z; t = f - z * per_z; }
    int tile_m, tile_n;
    grouped_tile(xcd_remap(t, per_z), tiles_m, tiles_n, p.group_m, tile_m, tile_n);
    m0 = tile_m * BM2; n0 = tile_n * 256;
    zo = z / p.d.batch_inner; zi = z - zo * p.d.batch_inner;
    Abase = (const unsigned char*)((const bf16_t*)p.d.A + ((int64_t)zo * p.d.a_zo + (int64_t)zi * p.d.a_zi));
    Bbase = (const unsigned char*)((const bf16_t*)p.d.B + ((int64_t)zo * p.d.b_zo + (int64_t)zi * p.d.b_zi));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (wave * 4 + i) * 8 + (lane >> 3);
      int ra = m0 + row; if (ra > p.d.M - 1) ra = p.d.M - 1;        // ragged edge: re-read the last valid row (results unused)
      aoff[i] = (uint32_t)(row_off((uint32_t)ra, p.d.a_mb, p.d.a_bs, p.d.lda, p.divAmb) + ack[i]) * 2u;
      const int rb = n0 + row;
      boff[i] = (uint32_t)((int64_t)rb * p.d.ldb + ack[i]) * 2u;
    }
  };
  auto issue = [&](int stage, int kt) {
    unsigned char* sA = smem + stage * STAGE + wave * 4096;
    unsigned char* sB = sA + TILE_A;
    const unsigned char* Ak = Abase + (size_t)kt * (BK * 2);
    const unsigned char* Bk = Bbase + (size_t)kt * (BK * 2);
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16((const bf16_t*)(Ak + aoff[i]), sA + i * 1024);
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16((const bf16_t*)(Bk + boff[i]), sB + i * 1024);
  };

  WaveEpilogue<KIND, V, ABL> epi(p);
  int f = blockIdx.x;
  setup(f);
  issue(0, 0);
  if (V) issue(1, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                    // K tile 0 (V: and 1) of the first tile is in LDS
  if (late) __builtin_amdgcn_s_barrier();          // stagger: waves 4-7 start one slot later
  __builtin_amdgcn_sched_barrier(0);
  int g = 0;                                       // K steps done so far by this workgroup (stage = g & 1)

  while (true) {
    const int cm0 = m0, cn0 = n0, czo = zo, czi = zi;        // the tile being accumulated
    const int ncol = cn0 + wn * 64 + li * 4;                 // this lane's 4 output columns in the epilogue
    const int fn = f + (int)gridDim.x;
    const bool has_next = fn < total;
    f32x4 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int b = 0; b < NT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // one K step.  MODE 0: stream K tile kt + 1 in R0, wait for it at the end of R1.  MODE 1 (V = 1, a tile's first step): K tile 1
    // is already there or in flight behind a wait that has been done; nothing to issue, nothing to wait for.  MODE 2 (the tile's
    // last step, the "seam"): R0 streams the NEXT tile's first K tile and loads this tile's bias / column scale.
    auto kstep = [&](auto mode_tag, int kt) {
      constexpr int MODE = decltype(mode_tag)::value;
      const unsigned char* Ah = smem + (g & 1) * STAGE;
      const unsigned char* Bh = Ah + TILE_A;
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
        // ---- R phase ----
        if (kh == 0) {
          if (MODE == 0) issue((g + 1) & 1, kt + 1);
          if (MODE == 2) {
            epi.load_cols(ncol, czo, czi);
            if (has_next) { setup(fn); issue((g + 1) & 1, 0); }
          }
        }
        u32x4 fa[MT], fb[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) fb[t] = *(const u32x4*)(Bh + lds_off(wn * 64 + t * 16 + li, kh * 4 + lg));
#pragma unroll
        for (int t = 0; t < MT; ++t) fa[t] = *(const u32x4*)(Ah + lds_off(wm * 128 + t * 16 + li, kh * 4 + lg));
        if (kh == 1 && MODE != 1) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // the streamed K tile (own pieces) landed
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- C phase ----
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
          for (int b = 0; b < NT; ++b) acc[a][b] = mfma16(fa[a], fb[b], acc[a][b]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
      ++g;
    };
    int kt = 0;
    if (V) { kstep(std::integral_constant<int, 1>{}, 0); kt = 1; }
    for (; kt + 1 < nk; ++kt) kstep(std::integral_constant<int, 0>{}, kt);
    kstep(std::integral_constant<int, 2>{}, nk - 1);

    if (!late) __builtin_amdgcn_s_barrier();                 // early half waits one slot: every wave is past its last stage read
    __builtin_amdgcn_sched_barrier(0);
    if (V && has_next) issue((g + 1) & 1, 1);                // the next tile's SECOND K tile, into the stage the seam step just read
    epi.compute(acc, scr, cm0 + wm * 128, ncol, czo, czi);
    if (V) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // that K tile and the epilogue's own loads: nothing but stores from here on
      __builtin_amdgcn_sched_barrier(0);
      epi.store_all(cm0 + wm * 128, ncol, czo, czi);
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (!has_next) break;
    f = fn;
    if (late) __builtin_amdgcn_s_barrier();                  // re-stagger for the next tile
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int KIND, int V, int ABL = 0>
int launch_pdma_k(const GemmP& p, hipStream_t stream) {
  constexpr int smem = 2 * (2 * BM2 * BK * 2) + 8 * 4096;    // two 64 KiB stages + 4 KiB of epilogue scratch per wave = 160 KiB
  static std::atomic<unsigned long long> done{0};
  if (int e = mtt_ensure_dyn_lds((const void*)gemm_pdma_kernel<KIND, V, ABL>, smem, done)) return e;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
    return (int)hipGetLastError();
  const int64_t total = (int64_t)((p.d.M + 255) / 256) * (p.d.N / 256) * p.d.batch;
  dim3 grid((unsigned)(total < cus ? total : cus), 1, 1);
  hipLaunchKernelGGL((gemm_pdma_kernel<KIND, V, ABL>), grid, dim3(512), smem, stream, p);
  return (int)hipGetLastError();
}
template <int V>
int launch_pdma(const GemmP& p, int kind, hipStream_t stream) {
  switch (kind) {
    case 0: return launch_pdma_k<0, V>(p, stream);
    case 1: return launch_pdma_k<1, V>(p, stream);
    case 2: return launch_pdma_k<2, V>(p, stream);
    case 3: return launch_pdma_k<3, V>(p, stream);
    case 4: return launch_pdma_k<4, V>(p, stream);
    default: return MTT_E_UNSUPPORTED;
  }
}

// ---------------------------------------------------------------------------------------------
// gemm_tn_kernel<CONVB>: D[m, n] = sum_k A[k, m] * B[k, n] with BOTH operands "row = reduction index" (MTT_OP_R: element (r, k) at
// base + k * ld + r) — the weight-gradient form dW = dy^T x on the token-major activations as they sit in HBM, no transposing copies
// (round 1 transposed both operands into reduction-contiguous buffers first: 26 ms of a 450 ms step) and, with CONVB, the 3x3 conv
// weight gradient (B = implicit im2col^T of x: n = (tap, ci), k = pixel).
//
// Same tile / wave / phase structure as gemm_dma_kernel<256> (256 x 256 x 64, 8 waves, LDS-DMA, staggered R / C phases).  What differs
// is the LDS image and the fragment read:
//   * a stage holds the A tile as [64 k][256 m] and the B tile as [64 k][256 n] (512-byte rows, exactly as in memory: every LDS-DMA
//     instruction moves two 512-byte row segments);
//   * MFMA fragments (lane (i, g) needs 8 consecutive k of column i) come from ds_read_b64_tr_b16, the LDS transpose read.  Measured
//     semantics on gfx950 (profiles/r02_probe_ds_read_b64_tr_b16.txt): inside a 16-lane group, lane r supplies an 8-byte address and
//     receives out[j] = element (r & 3) of the 8 bytes supplied by lane 4 j + (r >> 2).  With lane r pointing at row k0 + (r >> 2),
//     columns c0 + 4 (r & 3) .. + 3, lane r gets rows k0 .. k0 + 3 of column c0 + r: two such reads are one 16 x 32 fragment;
//   * bank conflicts: the 8 rows a 32-lane half touches are 512 bytes apart (same banks), so 32-byte units of a row are XOR-ed with
//     f(k) = (k & 3) | ((k >> 3) & 1) << 2 — applied to the per-lane SOURCE column of the DMA (the LDS image stays lane-linear) and
//     to the read address; f is a per-lane constant on the read side.
// ---------------------------------------------------------------------------------------------
MTT_DEV u32x2 ds_read_tr16(unsigned addr) {
  u32x2 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(addr) : "memory");
  return r;
}

template <bool CONVB>
__global__ __launch_bounds__(512, 1) void gemm_tn_kernel(const GemmP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int MT = 8, NT = 4, WAVES_N = 4;
  constexpr int TILE = BK * 256 * 2, STAGE = 2 * TILE;          // 32 KiB per operand tile, 64 KiB per stage
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int tiles_n = (p.d.N + 255) / 256, tiles_m = (p.d.M + 255) / 256;
  int tile_m, tile_n;
  grouped_tile(wg, tiles_m, tiles_n, p.group_m, tile_m, tile_n);
  const int m0 = tile_m * 256, n0 = tile_n * 256;
  const int z = blockIdx.z;
  const int zo = z / p.d.batch_inner, zi = z - zo * p.d.batch_inner;
  const bf16_t* Abase = (const bf16_t*)p.d.A + ((int64_t)zo * p.d.a_zo + (int64_t)zi * p.d.a_zi);
  const bf16_t* Bbase = (const bf16_t*)p.d.B + ((int64_t)zo * p.d.b_zo + (int64_t)zi * p.d.b_zi);

  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int late = wave >> 2;
  const int K = p.d.K;
  uint64_t zpage = (uint64_t)(uintptr_t)g_zero_page;
  asm volatile("" : "+s"(zpage));

  // ---- LDS-DMA source addressing: wave w moves k rows [8 w, 8 w + 8) of each tile as 4 pieces of 2 rows; lane -> (row, 16-byte chunk) ----
  const int c16 = lane & 31;                                    // 16-byte chunk of the 512-byte LDS row this lane fills
  int64_t a_col, b_col;                                         // element offset of this lane's source chunk inside a row (or -1)
  int fk[4];                                                    // f(k) of the 4 rows this lane fills (one per piece)
  int b_tapmask = 0, b_shift = 0;                               // CONVB: validity of this lane's tap per pixel is computed per K step
  int conv_dy = 0, conv_dx = 0;
  {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = wave * 8 + 2 * i + (lane >> 5);
      fk[i] = (k & 3) | (((k >> 3) & 1) << 2);
    }
  }
  // the source column depends on the row through the swizzle: col(i) = ((unit' & 8) | ((unit' & 7) ^ fk[i])) * 16 + (c16 & 1) * 8
  const int unitp = c16 >> 1, hb = c16 & 1;
  auto src_col = [&](int i) { return (((unitp & 8) | ((unitp & 7) ^ fk[i])) * 2 + hb) * 8; };

  auto issue = [&](int stage, int kt) {
    unsigned char* sA = smem + stage * STAGE + wave * 4096;     // 8 rows x 512 B per wave
    unsigned char* sB = sA + TILE;
    const int k0 = kt * BK + wave * 8 + (lane >> 5);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = k0 + 2 * i;
      const int ca = m0 + src_col(i);
      const bool oka = k < K && ca < p.d.M;
      const uint64_t srca = (uint64_t)(uintptr_t)(Abase + ((int64_t)k * p.d.lda + ca));
      glds16((const bf16_t*)(uintptr_t)(oka ? srca : zpage), sA + i * 1024);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = k0 + 2 * i;
      const int cb = n0 + src_col(i);
      bool okb = k < K && cb < p.d.N;
      uint64_t srcb;
      if constexpr (CONVB) {
        // n = tap * Cp + ci (chunks never straddle taps: Cp % 8 == 0); k = pixel inside this batch slice (slices are whole images)
        const uint32_t cbu = okb ? (uint32_t)cb : 0u;
        const uint32_t tap = fdiv(cbu, p.divCp);
        const int ci = (int)cbu - (int)tap * p.d.conv.Cp;
        const int ty = (int)fdiv(tap, p.div3), tx = (int)tap - 3 * ty;
        const int dy = (ty - 1) * p.d.conv.dil, dx = (tx - 1) * p.d.conv.dil;
        const uint32_t ku = k < K ? (uint32_t)k : 0u;
        const uint32_t t = fdiv(ku, p.divW);
        const int x = (int)(ku - t * (uint32_t)p.d.conv.W);
        const uint32_t bb = fdiv(t, p.divH);
        const int y = (int)(t - bb * (uint32_t)p.d.conv.H);
        okb = okb && ci < p.d.conv.C && (unsigned)(y + dy) < (unsigned)p.d.conv.H && (unsigned)(x + dx) < (unsigned)p.d.conv.W;
        srcb = (uint64_t)(uintptr_t)(Bbase + ((int64_t)((int)ku + dy * p.d.conv.W + dx) * p.d.ldb + ci));
      } else {
        srcb = (uint64_t)(uintptr_t)(Bbase + ((int64_t)k * p.d.ldb + cb));
      }
      glds16((const bf16_t*)(uintptr_t)(okb ? srcb : zpage), sB + i * 1024);
    }
  };

  // ---- transpose-read addressing (per-lane constants) ----
  const int r16 = lane & 15, g = lane >> 4;
  const int fr = (r16 >> 2) | ((g & 1) << 2);                   // f(k) of every row this lane reads
  const unsigned lbase = (unsigned)(uintptr_t)smem + (unsigned)((8 * g + (r16 >> 2)) * 512 + (r16 & 3) * 8);
  unsigned a_addr[MT], b_addr[NT];                              // stage 0, kh = 0, first 4-row half
#pragma unroll
  for (int t = 0; t < MT; ++t) a_addr[t] = lbase + (unsigned)((wm * 8 + (t ^ fr)) * 32);
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int unit = wn * 4 + t;
    b_addr[t] = lbase + (unsigned)TILE + (unsigned)(((unit & 8) | ((unit & 7) ^ fr)) * 32);
  }

  f32x4 acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = (K + BK - 1) / BK;
  issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (late) __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);

  for (int kt = 0; kt < nk; ++kt) {
    const unsigned st = (unsigned)((kt & 1) * STAGE);
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      // ---- R phase ----
      if (kh == 0 && kt + 1 < nk) issue((kt + 1) & 1, kt + 1);
      const unsigned off = st + (unsigned)(kh * 32 * 512);
      u32x2 bl[NT], bh[NT], al[MT], ah[MT];                  // asm loads: hipcc neither counts nor waits for them (guide §5.7)
#pragma unroll
      for (int t = 0; t < NT; ++t) { bl[t] = ds_read_tr16(b_addr[t] + off); bh[t] = ds_read_tr16(b_addr[t] + off + 4 * 512); }
#pragma unroll
      for (int t = 0; t < MT; ++t) { al[t] = ds_read_tr16(a_addr[t] + off); ah[t] = ds_read_tr16(a_addr[t] + off + 4 * 512); }
      // the wait names every destination as an in/out operand: no use (not even a register copy) can be scheduled above it
#define TNW(x) "+v"(x)
      if (kh == 1)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
                     : TNW(bl[0]), TNW(bl[1]), TNW(bl[2]), TNW(bl[3]), TNW(bh[0]), TNW(bh[1]), TNW(bh[2]), TNW(bh[3]),
                       TNW(al[0]), TNW(al[1]), TNW(al[2]), TNW(al[3]), TNW(al[4]), TNW(al[5]), TNW(al[6]), TNW(al[7]),
                       TNW(ah[0]), TNW(ah[1]), TNW(ah[2]), TNW(ah[3]), TNW(ah[4]), TNW(ah[5]), TNW(ah[6]), TNW(ah[7]) :: "memory");
      else
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : TNW(bl[0]), TNW(bl[1]), TNW(bl[2]), TNW(bl[3]), TNW(bh[0]), TNW(bh[1]), TNW(bh[2]), TNW(bh[3]),
                       TNW(al[0]), TNW(al[1]), TNW(al[2]), TNW(al[3]), TNW(al[4]), TNW(al[5]), TNW(al[6]), TNW(al[7]),
                       TNW(ah[0]), TNW(ah[1]), TNW(ah[2]), TNW(ah[3]), TNW(ah[4]), TNW(ah[5]), TNW(ah[6]), TNW(ah[7]) :: "memory");
#undef TNW
      u32x4 fa[MT], fb[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) fb[t] = (u32x4){bl[t][0], bl[t][1], bh[t][0], bh[t][1]};
#pragma unroll
      for (int t = 0; t < MT; ++t) fa[t] = (u32x4){al[t][0], al[t][1], ah[t][0], ah[t][1]};
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // ---- C phase ----
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = mfma16(fa[a], fb[b], acc[a][b]);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (!late) __builtin_amdgcn_s_barrier();
  __syncthreads();
  gemm_epilogue_auto<256, 2, WAVES_N, MT, NT>(p, acc, smem, m0, n0, zo, zi);
}

template <bool CONVB>
int launch_tn(const GemmP& p, hipStream_t stream) {
  constexpr int smem = 2 * 2 * BK * 256 * 2;
  static std::atomic<unsigned long long> done{0};
  if (int e = mtt_ensure_dyn_lds((const void*)gemm_tn_kernel<CONVB>, smem, done)) return e;
  const int tm = (p.d.M + 255) / 256, tn = (p.d.N + 255) / 256;
  dim3 grid(tm * tn, 1, p.d.batch);
  hipLaunchKernelGGL((gemm_tn_kernel<CONVB>), grid, dim3(512), smem, stream, p);
  return (int)hipGetLastError();
}

FastDiv make_div(uint32_t dv) {
  FastDiv f; f.d = dv ? dv : 1u;
  uint32_t s = 0; while ((1ull << s) < f.d) ++s;
  f.shift = s;
  f.magic = (uint32_t)((((1ull << 32) * ((1ull << s) - f.d)) / f.d) + 1ull);
  return f;
}

template <int AOP, int BOP, int MODE>
int launch(const GemmP& p, hipStream_t stream) {
  constexpr int smem = TILE_BYTES * 2 * (MODE == 2 ? 2 : 1) * 2;
  static std::atomic<unsigned long long> done{0};
  if (int e = mtt_ensure_dyn_lds((const void*)gemm_kernel<AOP, BOP, MODE>, smem, done)) return e;
  dim3 grid(p.tiles_m * p.tiles_n, 1, p.d.batch);
  hipLaunchKernelGGL((gemm_kernel<AOP, BOP, MODE>), grid, dim3(256), smem, stream, p);
  return (int)hipGetLastError();
}

}  // namespace

extern "C" int mtt_abi_version(void) { return MTT_ABI_VERSION; }

// sizeof() of each descriptor, so a foreign-language binding can verify its struct mirror at load time
extern "C" size_t mtt_desc_size(int which) {
  switch (which) {
    case 0: return sizeof(mtt_gemm_desc);
    case 1: return sizeof(mtt_attn_desc);
    case 2: return sizeof(mtt_softmax_desc);
    case 3: return sizeof(mtt_ln_desc);
    case 4: return sizeof(mtt_chanlogit_desc);
    case 5: return sizeof(mtt_modulate_desc);
    case 6: return sizeof(mtt_ctr_desc);
    case 7: return sizeof(mtt_resize_desc);
    case 8: return sizeof(mtt_bn_desc);
    case 9: return sizeof(mtt_conv_geom);
    case 10: return sizeof(mtt_dwconv_desc);
    case 11: return sizeof(mtt_pool_desc);
    case 12: return sizeof(mtt_lnmt_desc);
    case 13: return sizeof(mtt_attnmsg_desc);
    case 14: return sizeof(mtt_convt_desc);
    case 15: return sizeof(mtt_adam_desc);
    case 16: return sizeof(mtt_loss_desc);
    case 17: return sizeof(mtt_upconv_desc);
    case 18: return sizeof(mtt_gather_desc);
    case 19: return sizeof(mtt_winattn_desc);
    case 20: return sizeof(mtt_chanattn_desc);
    case 21: return sizeof(mtt_conv3s2_desc);
    default: return 0;
  }
}

// Kernel choice, a pure function of the descriptor.  Return codes (also what mtt_gemm_variant reports):
//   0 register-staged 128 x 128 (general: any operand layout / dtype / precision)
//   1 LDS-DMA 128 x 128, 4-stage ring (round-1 kernel; forced only)
//   3 LDS-DMA 256 x 256 phased / staggered (gemm_dma_kernel<256>)      4 the same with a 256 x 128 tile (gemm_dma_kernel<128>)
//   5 LDS-DMA 256 x 256 lock-step 2-stage (round-1 kernel; forced only, kept for A/B measurements)
//   6 token-major weight-gradient kernel (gemm_tn_kernel): LDS-DMA + ds_read_b64_tr_b16 fragments
//   7 persistent LDS-DMA 256 x 256 (gemm_pdma_kernel): plain bf16 operands, N % 256 == 0, one of the specialised epilogue kinds
// d.variant = MTT_GEMM_AUTO applies the policy; another value forces that kernel where it is applicable.
// K % 64 == 0 and 32-bit per-lane byte offsets from the batch member's base: the fast source addressing of the 256 x 256 LDS-DMA kernels
static bool dma_fastaddr_ok(const mtt_gemm_desc& d) {
  if (d.K % 64) return false;
  const int64_t a_span = (d.a_mb > 0 ? (int64_t)((d.M - 1) / d.a_mb) * d.a_bs + (int64_t)((d.M - 1) % d.a_mb) * d.lda : (int64_t)(d.M - 1) * d.lda) + d.K;
  const int64_t b_span = (int64_t)(d.N - 1) * d.ldb + d.K;
  return a_span < (1ll << 31) && b_span < (1ll << 31);
}
static bool pdma_eligible(const mtt_gemm_desc& d) {
  if ((d.N % 256) || (d.K % 64) || d.K < 128 || d.n_store > d.N || epilogue_kind_of(d) < 0) return false;
  // 32-bit per-lane byte offsets from the batch member's base
  const int64_t a_span = (d.a_mb > 0 ? (int64_t)((d.M - 1) / d.a_mb) * d.a_bs + (int64_t)((d.M - 1) % d.a_mb) * d.lda : (int64_t)(d.M - 1) * d.lda) + d.K;
  const int64_t b_span = (int64_t)(d.N - 1) * d.ldb + d.K;
  return a_span < (1ll << 31) && b_span < (1ll << 31);
}
// AUTO policy switch for the persistent kernel (set after the A/B on MI355X: tools/gemm_bench.py, profiles/r02_gemm_bench_m_*)
constexpr bool PDMA_BY_DEFAULT = false;
// AUTO policy switch for the swapped-MFMA / direct-store epilogue form of the 256 x 256 kernel (set after the A/B on MI355X)
constexpr bool DIRECT_EPILOGUE_BY_DEFAULT = false;
static int gemm_variant_for(const mtt_gemm_desc& d) {
  // 6: token-major weight-gradient kernel (gemm_tn_kernel): both operands MTT_OP_R (B may be the implicit im2col^T), bf16
  const bool tn = d.prec == MTT_PREC_BF16 && d.a_op == MTT_OP_R && (d.b_op == MTT_OP_R || d.b_op == MTT_OP_CONV_R) &&
                  d.a_dtype == MTT_BF16 && d.b_dtype == MTT_BF16 && d.store_mode == MTT_STORE_ROWS;
  if (tn && d.variant != MTT_GEMM_GENERAL) {
    if (d.variant == MTT_GEMM_DMA256) return 6;
    const int64_t pm = (d.M + 255) / 256 * 256, pn = (d.N + 255) / 256 * 256;
    const int batch = d.batch < 1 ? 1 : d.batch;
    const bool fills = (pm / 256) * (pn / 256) * batch >= 64 && d.K >= 512;
    // 256-wide tiles must not waste more than ~40 % of the MFMA work (narrow decoder outputs stay on the 128-wide general kernel)
    if (fills && 100 * (int64_t)d.M * d.N >= 60 * pm * pn) return 6;
    return 0;
  }
  const bool conv = d.a_op == MTT_OP_CONV_K;
  const bool dma = d.prec == MTT_PREC_BF16 && (d.a_op == MTT_OP_K || conv) && d.b_op == MTT_OP_K && (d.K % 8) == 0 &&
                   d.a_dtype == MTT_BF16 && d.b_dtype == MTT_BF16;
  if (!dma || d.variant == MTT_GEMM_GENERAL) return 0;
  const bool v1_ok = !conv && (d.K % BK) == 0;
  if (d.variant == MTT_GEMM_DMA128 && v1_ok) return 1;
  if (d.variant == MTT_GEMM_DMA256_V1 && v1_ok) return 5;
  const int n256 = (d.N + 255) / 256 * 256, n128 = (d.N + 127) / 128 * 128;
  const int bn = 100 * n128 <= 85 * n256 ? 128 : 256;   // the narrower tile only where it saves >= 15 % of the columns (N = 300, 350, 576 ...)
  if (d.variant == MTT_GEMM_DMA256 || d.variant == MTT_GEMM_DMA256_S1) return bn == 256 ? 3 : 4;
  if (d.variant == MTT_GEMM_ABLATE_NO_EPILOGUE || d.variant == MTT_GEMM_ABLATE_NO_KLOOP || d.variant == MTT_GEMM_DMA256_SKEW ||
      d.variant == MTT_GEMM_ABLATE_NO_STORES || d.variant == MTT_GEMM_ABLATE_NO_STAGING) return 3;
  // AUTO.  Measured on MI355X (profiles/r02_gemm_bench_b*.log, r02_conv_bench_b.log, B = 63 shapes): the 256 x 256 DMA tile wins for
  // wide outputs (qkv / proj / fc1 / fc2: 830-1170 vs 610-740 TFLOP/s on the register-staged 128 x 128 kernel), but the 256 x 128 DMA
  // tile LOSES to it on the narrow decoder shapes (N = 300 / 350: 280-330 vs 300-370) and on the implicit-GEMM 3x3 conv
  // (510 vs 640): with half the MFMAs per K step the 6 LDS-DMA pieces a wave issues (+ the im2col address math) are no longer
  // covered.  So: DMA kernel for plain GEMMs with N >= 512 columns of 256-wide tiles, the general kernel otherwise.
  const int batch = d.batch < 1 ? 1 : d.batch;
  const int64_t blocks = (int64_t)((d.M + 255) / 256) * ((d.N + 255) / 256) * batch;
  // MTT_GEMM_DMA256_PERSIST(_STAG) = this policy with the persistent kernel wherever it is eligible (so that a whole training step can
  // be A/B-ed by forcing one variant value on every mtt_gemm call: bench.py --gemm-variant)
  const bool want_p = (PDMA_BY_DEFAULT && d.variant != MTT_GEMM_DMA256_NONPERSIST) || d.variant == MTT_GEMM_DMA256_PERSIST || d.variant == MTT_GEMM_DMA256_PERSIST_V0 ||
                      (d.variant >= MTT_GEMM_PDMA_ABLATE_NO_EPILOGUE && d.variant <= MTT_GEMM_PDMA_ABLATE_NO_BIAS);
  if (!conv && bn == 256 && d.M >= 512 && d.N >= 512 && blocks >= 96) return (want_p && pdma_eligible(d)) ? 7 : 3;
  return 0;
}
extern "C" int mtt_gemm_variant(const mtt_gemm_desc* d) { return d ? gemm_variant_for(*d) : MTT_E_BADARG; }

extern "C" int mtt_gemm(const mtt_gemm_desc* dd, void* stream) {
  if (!dd || !dd->A || !dd->B || !dd->D) return MTT_E_BADARG;
  GemmP p; p.d = *dd;
  mtt_gemm_desc& d = p.d;
  if (d.M <= 0 || d.N <= 0 || d.K <= 0) return MTT_E_BADARG;
  if (d.batch < 1) d.batch = 1;
  if (d.batch_inner < 1) d.batch_inner = 1;
  const bool a_k = d.a_op == MTT_OP_K || d.a_op == MTT_OP_CONV_K, b_k = d.b_op == MTT_OP_K;
  const int Kp8 = (d.K + 7) / 8 * 8;
  if ((a_k && d.a_op == MTT_OP_K && d.lda < Kp8) || (b_k && d.ldb < Kp8)) return MTT_E_ALIGN;
  if (d.a_op == MTT_OP_CONV_K && (d.K % 8)) return MTT_E_ALIGN;
  if ((d.lda % 8) || (d.ldb % 8)) return MTT_E_ALIGN;
  if (((uintptr_t)d.A & 15) || ((uintptr_t)d.B & 15)) return MTT_E_ALIGN;
  if (d.prec == MTT_PREC_X3 && (d.a_dtype != MTT_F32 || d.b_dtype != MTT_F32)) return MTT_E_UNSUPPORTED;
  if ((d.aux_in || d.aux_out) && (d.ldaux <= 0 || (d.ldaux % 8))) return MTT_E_BADARG;
  if ((d.ldd % 8) || ((uintptr_t)d.D & 15) || (d.d_bs % 8) || (d.d_zo % 8) || (d.d_zi % 8)) return MTT_E_ALIGN;
  if (d.resid && ((d.ldr % 4) || ((uintptr_t)d.resid & 15) || (d.r_bs % 4))) return MTT_E_ALIGN;
  const bool conv = d.a_op == MTT_OP_CONV_K || d.b_op == MTT_OP_CONV_R;
  if (conv) {
    if (d.conv.H <= 0 || d.conv.W <= 0 || d.conv.Cp % 8 || d.conv.C > d.conv.Cp || d.conv.dil < 1) return MTT_E_BADARG;
    if (d.a_op == MTT_OP_CONV_K && d.K != 9 * d.conv.Cp) return MTT_E_BADARG;
    if (d.b_op == MTT_OP_CONV_R && d.N != 9 * d.conv.Cp) return MTT_E_BADARG;
  }
  if (d.store_mode == MTT_STORE_PIXSHUF2 && (d.ps_H <= 0 || d.ps_W <= 0 || d.ps_Co <= 0 || d.N != 4 * d.ps_Co)) return MTT_E_BADARG;
  p.divW = make_div(conv ? d.conv.W : 1); p.divH = make_div(conv ? d.conv.H : 1);
  p.divCp = make_div(conv ? d.conv.Cp : 1); p.div3 = make_div(3);
  p.divAmb = make_div(d.a_mb > 0 ? d.a_mb : 1); p.divDmb = make_div(d.d_mb > 0 ? d.d_mb : 1);
  p.divRmb = make_div(d.r_mb > 0 ? d.r_mb : 1);
  p.divPsW = make_div(d.ps_W > 0 ? d.ps_W : 1); p.divPsH = make_div(d.ps_H > 0 ? d.ps_H : 1);
  p.divPsCo = make_div(d.ps_Co > 0 ? d.ps_Co : 1);
  p.tiles_m = (d.M + BM - 1) / BM; p.tiles_n = (d.N + BN - 1) / BN;
  p.group_m = 4;
  hipStream_t s = (hipStream_t)stream;
  int mode;
  if (d.prec == MTT_PREC_X3) mode = 2;
  else if (d.b_dtype != MTT_BF16) return MTT_E_UNSUPPORTED;      /* bf16 mode: B must be bf16 (A may be f32) */
  else mode = d.a_dtype == MTT_F32 ? 1 : 0;
  if (mode == 0) {
    const int v = gemm_variant_for(d);
    const bool conv_a = d.a_op == MTT_OP_CONV_K;
    // LDS-DMA schedule: 0 (one burst in R0) by default — measured equal or better than the balanced schedule on 4 of 5 shapes
    // (profiles/r02_gemm_bench_c_dma_schedules.log: qkv 871 / 869, proj 896 / 889, fc2 1171 / 1157, 8192^3 1326 / 1272 TFLOP/s;
    // fc1+GELU 789 / 817) and it is the one the full-size parity tests ran on; MTT_GEMM_DMA256_S1 forces the balanced one
    if (v == 3 && !conv_a && d.variant == MTT_GEMM_ABLATE_NO_EPILOGUE) return launch_dma<256, false, 2>(p, s);
    if (v == 3 && !conv_a && d.variant == MTT_GEMM_ABLATE_NO_KLOOP) return launch_dma<256, false, 3>(p, s);
    if (v == 3 && !conv_a && d.variant == MTT_GEMM_DMA256_SKEW) return launch_dma<256, false, 4>(p, s);
    if (v == 3 && !conv_a && d.variant == MTT_GEMM_ABLATE_NO_STORES) return launch_dma<256, false, 5>(p, s);
    if (v == 3 && !conv_a && d.variant == MTT_GEMM_ABLATE_NO_STAGING) return launch_dma<256, false, 6>(p, s);
    if (v == 3 && !conv_a && d.variant != MTT_GEMM_DMA256_S1 && d.variant != MTT_GEMM_DMA256_SLOWADDR && dma_fastaddr_ok(d))
      return (DIRECT_EPILOGUE_BY_DEFAULT ? d.variant != MTT_GEMM_DMA256_LDS_EPILOGUE : d.variant == MTT_GEMM_DMA256_DIRECT)
                 ? launch_dma<256, false, 8>(p, s) : launch_dma<256, false, 7>(p, s);
    if (v == 3) return conv_a ? launch_dma<256, true, 0>(p, s) : (d.variant == MTT_GEMM_DMA256_S1 ? launch_dma<256, false, 1>(p, s) : launch_dma<256, false, 0>(p, s));
    if (v == 4) return conv_a ? launch_dma<128, true, 0>(p, s) : launch_dma<128, false, 0>(p, s);
    if (v == 7 && epilogue_kind_of(d) == 0 && d.variant >= MTT_GEMM_PDMA_ABLATE_NO_EPILOGUE && d.variant <= MTT_GEMM_PDMA_ABLATE_NO_BIAS)
      return d.variant == MTT_GEMM_PDMA_ABLATE_NO_EPILOGUE ? launch_pdma_k<0, 1, 1>(p, s)
             : (d.variant == MTT_GEMM_PDMA_ABLATE_NO_STORES ? launch_pdma_k<0, 1, 2>(p, s) : launch_pdma_k<0, 1, 3>(p, s));
    if (v == 7) return d.variant == MTT_GEMM_DMA256_PERSIST_V0 ? launch_pdma<0>(p, epilogue_kind_of(d), s) : launch_pdma<1>(p, epilogue_kind_of(d), s);
    if (v == 6) return d.b_op == MTT_OP_CONV_R ? launch_tn<true>(p, s) : launch_tn<false>(p, s);
    if (v == 5) return launch_fast256(p, s);
    if (v == 1) return launch_fast(p, s);
  }
#define MTT_CASE(AO, BO) \
  if (d.a_op == AO && d.b_op == BO) \
    return mode == 2 ? launch<AO, BO, 2>(p, s) : (mode == 1 ? launch<AO, BO, 1>(p, s) : launch<AO, BO, 0>(p, s));
  MTT_CASE(MTT_OP_K, MTT_OP_K)
  MTT_CASE(MTT_OP_K, MTT_OP_R)
  MTT_CASE(MTT_OP_R, MTT_OP_R)
  MTT_CASE(MTT_OP_CONV_K, MTT_OP_K)
  MTT_CASE(MTT_OP_R, MTT_OP_CONV_R)
#undef MTT_CASE
  return MTT_E_UNSUPPORTED;
}
