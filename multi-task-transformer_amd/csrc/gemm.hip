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

#ifndef MTT_GROUP_M
#define MTT_GROUP_M 4        // tile rows swept together by the grouped tile order (tools/gemm_bench.py measures other values on library builds)
#endif
#ifndef MTT_RING
#define MTT_RING 1           // 1: gemm_ring3_kernel takes the split-plane (x3) LDS-DMA calls; 0: round 3's K-concatenated gemm_dma_kernel<2> (A/B builds)
#endif
constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;   // 16 KiB per bf16 plane

struct GemmP {
  mtt_gemm_desc d;
  FastDiv divW, divH, divCp, div3;   // conv geometry
  FastDiv divAmb, divDmb, divRmb;    // row-group mappings
  FastDiv divPsW, divPsH, divPsCo;   // pixel-shuffle store
  int tiles_m, tiles_n;
  int group_m;
#ifdef MTT_GEMM_TRACE
  unsigned long long* trace;       // experiment builds only (tools/gemm_trace.py): 8 words per workgroup
#endif
};

#ifdef MTT_GEMM_TRACE
// experiment builds only: per-workgroup timestamps of the LDS-DMA kernels (thread 0): word 0 s_memrealtime at entry, 1..4 s_memtime at
// entry / after the prologue / after the K loop / after the epilogue, 5 s_memrealtime at exit, 6 HW_ID | XCC_ID << 32, 7 tile index
MTT_DEV unsigned long long trace_hwid() {
  unsigned a, b;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(a));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(b));
  return (unsigned long long)a | ((unsigned long long)b << 32);
}
#define MTT_TRACE(slot) do { if (p.trace && threadIdx.x == 0) p.trace[(size_t)(blockIdx.x + gridDim.x * blockIdx.z) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#define MTT_TRACE_RT(slot) do { if (p.trace && threadIdx.x == 0) p.trace[(size_t)(blockIdx.x + gridDim.x * blockIdx.z) * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define MTT_TRACE_ID(tile) do { if (p.trace && threadIdx.x == 0) { p.trace[(size_t)(blockIdx.x + gridDim.x * blockIdx.z) * 8 + 6] = trace_hwid(); p.trace[(size_t)(blockIdx.x + gridDim.x * blockIdx.z) * 8 + 7] = (tile); } } while (0)
#else
#define MTT_TRACE(slot) do {} while (0)
#define MTT_TRACE_RT(slot) do {} while (0)
#define MTT_TRACE_ID(tile) do {} while (0)
#endif

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

// lo plane of 8 values whose hi plane (packed bf16) is `hi`: bf16(v - float(hi))
MTT_DEV u32x4 split_lo(const float (&v)[8], u32x4 hi) {
  return (u32x4){pack2(v[0] - lo_of(hi.x), v[1] - hi_of(hi.x)), pack2(v[2] - lo_of(hi.y), v[3] - hi_of(hi.y)),
                 pack2(v[4] - lo_of(hi.z), v[5] - hi_of(hi.z)), pack2(v[6] - lo_of(hi.w), v[7] - hi_of(hi.w))};
}

// ---------------------------------------------------------------------------------------------
// Column sums of the stored tile (mtt_gemm_desc.colsum_out): every thread of the epilogue owns one 8-column chunk of the tile
// (tid = row_slot * CHUNKS + chunk) and has summed its rows into cacc[8]; this adds the row slots of the workgroup in a fixed order
// (lanes of a wave by xor-shuffles, waves in index order through LDS) and writes the tile's partial row of the workspace.
template <int NTHREADS, int TBN>
MTT_DEV void epilogue_colsum_flush(float (&cacc)[8], float* red, float* part, int ncols) {
  constexpr int CHUNKS = TBN / 8, W = NTHREADS / 64;
  static_assert(CHUNKS <= 64 && 64 % CHUNKS == 0, "tile width");
#pragma unroll
  for (int o = CHUNKS; o < 64; o <<= 1)
#pragma unroll
    for (int j = 0; j < 8; ++j) cacc[j] += __shfl_xor(cacc[j], o, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane < CHUNKS) {
#pragma unroll
    for (int j = 0; j < 8; ++j) red[wave * TBN + lane * 8 + j] = cacc[j];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < TBN; c += NTHREADS) {
    float t = red[c];
#pragma unroll
    for (int w = 1; w < W; ++w) t += red[w * TBN + c];
    if (c < ncols) part[c] = t;
  }
}
MTT_DEV float stored_value(float v, int d_dtype) { return d_dtype == MTT_BF16 ? bf2f(f2bf(v)) : v; }

// Shared epilogue.  Block tile TBN columns wide, WAVES_M x WAVES_N waves, each wave MT x NTL tiles of 16 x 16:
//   acc[a][b][r] = D[(wm*MT + a)*16 + lg*4 + r][(wn*NTL + b)*16 + li]
// ---------------------------------------------------------------------------------------------
template <int TBN, int WAVES_M, int WAVES_N, int MT, int NTL, bool CS = true>
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
  // colsum_out: this thread's column sums over its rows of the tile, kept in its own 8 LDS slots behind the staging area (this
  // general path is at the register limit in some instantiations; the specialised path below keeps them in registers)
#define csl (ep + 64 * EP_LD + (threadIdx.x << 3))
  if (CS && d.colsum_out) {
#pragma unroll
    for (int j = 0; j < 8; ++j) csl[j] = 0.f;
  }

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
          ep[((gt & 3) * 16 + lg * 4 + r) * EP_LD + (wn * NTL + b) * 16 + li] = acc[a][b][r];
        }
    }
    __syncthreads();
    if (ncol0 < n_store || (d.store_mode == MTT_STORE_PIXSHUF2 && ncol0 < d.N)) {
#pragma unroll 1
      for (int i = 0; i < 64 / RPP; ++i) {
        const int rl = threadIdx.x / CHUNKS + RPP * i;
        const int m = m0 + half * 64 + rl;
        if (m >= d.M) continue;
        float v[8];
        {
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
          if (d.act == MTT_ACT_GELU_DAUX) {                    // aux_out = GELU'(z), D = GELU(z)
            float a8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) gelu_both_f(v[j], v[j], a8[j]);
            if (d.aux_dtype == MTT_F32) {
              *(float4*)((float*)d.aux_out + auxoff) = make_float4(a8[0], a8[1], a8[2], a8[3]);
              *(float4*)((float*)d.aux_out + auxoff + 4) = make_float4(a8[4], a8[5], a8[6], a8[7]);
            } else {
              *(u32x4*)((bf16_t*)d.aux_out + auxoff) = (u32x4){pack2(a8[0], a8[1]), pack2(a8[2], a8[3]), pack2(a8[4], a8[5]), pack2(a8[6], a8[7])};
            }
          } else if (d.aux_out) {
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
          } else if (d.act == MTT_ACT_MUL_AUX) {
            if (d.aux_dtype == MTT_F32) {
              const float4 z0 = *(const float4*)((const float*)d.aux_in + auxoff);
              const float4 z1 = *(const float4*)((const float*)d.aux_in + auxoff + 4);
              v[0] *= z0.x; v[1] *= z0.y; v[2] *= z0.z; v[3] *= z0.w; v[4] *= z1.x; v[5] *= z1.y; v[6] *= z1.z; v[7] *= z1.w;
            } else {
              const u32x4 u = *(const u32x4*)((const bf16_t*)d.aux_in + auxoff);
#pragma unroll
              for (int t = 0; t < 4; ++t) { v[2 * t] *= lo_of(u[t]); v[2 * t + 1] *= hi_of(u[t]); }
            }
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
          if (CS && d.colsum_out) {
#pragma unroll
            for (int j = 0; j < 8; ++j) csl[j] += stored_value(v[j], d.d_dtype);
          }
          if (d.d_dtype == MTT_F32) {
            *(float4*)((float*)d.D + doff) = make_float4(v[0], v[1], v[2], v[3]);
            *(float4*)((float*)d.D + doff + 4) = make_float4(v[4], v[5], v[6], v[7]);
          } else {
            const u32x4 hi = (u32x4){pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
            *(u32x4*)((bf16_t*)d.D + doff) = hi;
            if (d.d_dtype == MTT_SPLIT) *(u32x4*)((bf16_t*)d.D_lo + doff) = split_lo(v, hi);
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
              if (d.act == MTT_ACT_GELU_DAUX) {
                float dg;
                gelu_both_f(w, w, dg);
                st_elem(d.aux_out, auxoff + j, d.aux_dtype, dg);
              } else if (d.aux_out) st_elem(d.aux_out, auxoff + j, d.aux_dtype, w);
              if (d.act == MTT_ACT_GELU) w = gelu_f(w);
              else if (d.act == MTT_ACT_MUL_AUX) w *= ld_elem(d.aux_in, auxoff + j, d.aux_dtype);
              else if (d.act == MTT_ACT_RELU) w = fmaxf(w, 0.0f);
              else if (d.act == MTT_ACT_GELU_BWD) w *= gelu_grad_f(ld_elem(d.aux_in, auxoff + j, d.aux_dtype));
              else if (d.act == MTT_ACT_RELU_BWD) w = ld_elem(d.aux_in, auxoff + j, d.aux_dtype) > 0.0f ? w : 0.0f;
              w *= rs;
              if (d.resid) w += d.resid[roff + j];
              if (CS && d.colsum_out) csl[j] += stored_value(w, d.d_dtype);
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
              if (d.d_dtype == MTT_SPLIT) {
                const bf16_t h = f2bf(w);
                ((bf16_t*)d.D)[doff + j] = h;
                ((bf16_t*)d.D_lo)[doff + j] = f2bf(w - bf2f(h));
              } else st_elem(d.D, doff + j, d.d_dtype, w);
            }
          }
        }
      }
    }
    __syncthreads();
  }
  if (CS && d.colsum_out) {
    constexpr int TBM = WAVES_M * MT * 16;
    float cacc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) cacc[j] = csl[j];
    epilogue_colsum_flush<NTHREADS, TBN>(cacc, ep, d.colsum_ws + (int64_t)(m0 / TBM) * ((d.N + 7) / 8 * 8) + n0, d.N - n0);
  }
#undef csl
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
//   5  D split (hi / lo bf16 planes) = acc + bias        (x3-forward mode: outputs that feed the next LDS-DMA GEMM)
//   6  D split = GELU(z), z = acc + bias; aux_out bf16 = z when given   (fc1 of the x3-forward mode)
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
    if (d.resid || d.rowscale) return -1;
    return d.d_dtype == MTT_SPLIT ? 5 : 0;
  }
  const bool gelu = d.act == MTT_ACT_GELU || (d.act == MTT_ACT_GELU_DAUX && auxo);
  if (gelu && d.d_dtype == MTT_BF16 && !d.resid && !d.rowscale && !auxi) return 2;
  if (gelu && d.d_dtype == MTT_SPLIT && !d.resid && !d.rowscale && !auxi) return 6;
  if ((d.act == MTT_ACT_GELU_BWD || d.act == MTT_ACT_MUL_AUX) && d.d_dtype == MTT_BF16 && !d.resid && !d.rowscale && auxi && !auxo) return 4;
  return -1;
}
MTT_DEV int fast_epilogue_kind(const mtt_gemm_desc& d, int m0, int n0, int tbm, int tbn) {
  if (m0 + tbm > d.M || n0 + tbn > d.N) return -1;                                   // interior tiles only
  const int k = epilogue_kind_of(d);
  return (d.colsum_out && k != 0 && k != 4) ? -1 : k;                                // column sums: specialised for the bf16 gradient kinds
}

template <int KIND, int TBN, int WAVES_M, int WAVES_N, int MT, int NTL, bool CS = true>
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
  constexpr bool CSUM = CS && (KIND == 0 || KIND == 4);        // bf16 gradients: the kinds whose column sums are a bias gradient
  float cacc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

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
      if (KIND == 2 || KIND == 6) {
        if (d.act == MTT_ACT_GELU_DAUX) {                       // (workgroup-uniform) aux_out = GELU'(z) for a one-multiply backward epilogue
          float a8[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) gelu_both_f(v[j], v[j], a8[j]);
          *(u32x4*)((bf16_t*)d.aux_out + (zAux + m * d.ldaux)) = (u32x4){pack2(a8[0], a8[1]), pack2(a8[2], a8[3]), pack2(a8[4], a8[5]), pack2(a8[6], a8[7])};
        } else {
          if (d.aux_out)
            *(u32x4*)((bf16_t*)d.aux_out + (zAux + m * d.ldaux)) = (u32x4){pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = gelu_f(v[j]);
        }
      }
      if (KIND == 4) {
        const u32x4 u = za[i];
        if (d.act == MTT_ACT_MUL_AUX) {                         // (workgroup-uniform) aux_in = GELU'(z) from the forward
          v[0] *= lo_of(u.x); v[1] *= hi_of(u.x); v[2] *= lo_of(u.y); v[3] *= hi_of(u.y);
          v[4] *= lo_of(u.z); v[5] *= hi_of(u.z); v[6] *= lo_of(u.w); v[7] *= hi_of(u.w);
        } else {
          v[0] *= gelu_grad_f(lo_of(u.x)); v[1] *= gelu_grad_f(hi_of(u.x)); v[2] *= gelu_grad_f(lo_of(u.y)); v[3] *= gelu_grad_f(hi_of(u.y));
          v[4] *= gelu_grad_f(lo_of(u.z)); v[5] *= gelu_grad_f(hi_of(u.z)); v[6] *= gelu_grad_f(lo_of(u.w)); v[7] *= gelu_grad_f(hi_of(u.w));
        }
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
        const u32x4 hi = (u32x4){pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
        *(u32x4*)((bf16_t*)d.D + (zD + m * d.ldd)) = hi;
        if (KIND == 5 || KIND == 6) *(u32x4*)((bf16_t*)d.D_lo + (zD + m * d.ldd)) = split_lo(v, hi);
        if (CSUM && d.colsum_out) {                              // sums of the ROUNDED values (what a later read of D would see)
          cacc[0] += lo_of(hi.x); cacc[1] += hi_of(hi.x); cacc[2] += lo_of(hi.y); cacc[3] += hi_of(hi.y);
          cacc[4] += lo_of(hi.z); cacc[5] += hi_of(hi.z); cacc[6] += lo_of(hi.w); cacc[7] += hi_of(hi.w);
        }
      }
    }
    __syncthreads();
  }
  if (CSUM && d.colsum_out) {
    constexpr int TBM = WAVES_M * MT * 16;
    epilogue_colsum_flush<NTHREADS, TBN>(cacc, ep, d.colsum_ws + (int64_t)(m0 / TBM) * ((d.N + 7) / 8 * 8) + n0, TBN);
  }
}

// epilogue dispatch (workgroup-uniform): specialised path for interior tiles of the hot call sites, general path otherwise
template <int TBN, int WAVES_M, int WAVES_N, int MT, int NTL, bool CS = true>
MTT_DEV void gemm_epilogue_auto(const GemmP& p, f32x4 (&acc)[MT][NTL], unsigned char* smem, int m0, int n0, int zo, int zi) {
  const int kind = p.d.variant == MTT_GEMM_GENERAL_EPILOGUE ? -1 : fast_epilogue_kind(p.d, m0, n0, WAVES_M * MT * 16, TBN);
  if (kind == 0) gemm_epilogue_fast<0, TBN, WAVES_M, WAVES_N, MT, NTL, CS>(p, acc, smem, m0, n0, zo, zi);
  else if (kind == 1) gemm_epilogue_fast<1, TBN, WAVES_M, WAVES_N, MT, NTL, CS>(p, acc, smem, m0, n0, zo, zi);
  else if (kind == 2) gemm_epilogue_fast<2, TBN, WAVES_M, WAVES_N, MT, NTL, CS>(p, acc, smem, m0, n0, zo, zi);
  else if (kind == 3) gemm_epilogue_fast<3, TBN, WAVES_M, WAVES_N, MT, NTL, CS>(p, acc, smem, m0, n0, zo, zi);
  else if (kind == 4) gemm_epilogue_fast<4, TBN, WAVES_M, WAVES_N, MT, NTL, CS>(p, acc, smem, m0, n0, zo, zi);
  else if (kind == 5) gemm_epilogue_fast<5, TBN, WAVES_M, WAVES_N, MT, NTL, CS>(p, acc, smem, m0, n0, zo, zi);
  else if (kind == 6) gemm_epilogue_fast<6, TBN, WAVES_M, WAVES_N, MT, NTL, CS>(p, acc, smem, m0, n0, zo, zi);
  else gemm_epilogue<TBN, WAVES_M, WAVES_N, MT, NTL, CS>(p, acc, smem, m0, n0, zo, zi);
}

// ---------------------------------------------------------------------------------------------
// MODE: 0 = bf16 MFMA, A and B bf16;  1 = bf16 MFMA, A f32 (converted while staging), B bf16;  2 = X3 (both f32);
//       3 = bf16 MFMA, A and B f32 (both converted while staging);  4 = bf16 MFMA, A bf16, B f32
template <int AOP, int BOP, int MODE>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmP p) {
  constexpr bool X3 = MODE == 2, AF32 = MODE >= 1 && MODE <= 3, BF32 = MODE >= 2;
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

  gemm_epilogue_auto<BN, 2, 2, 4, 4, MODE != 3>(p, acc, smem, m0, n0, zo, zi);   // MODE 3 is at 256 VGPRs: no column-sum option there
}


MTT_DEV void glds16(const bf16_t* src, unsigned char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

constexpr int BM2 = 256;      // row tile of the LDS-DMA kernels

// ---------------------------------------------------------------------------------------------
// gemm_dma_kernel<ADDR>: the main kernel (every encoder Linear forward and input gradient).  256 x 256 x 64 tile, 8 waves (2 x 4, each
// 128 x 64 = 8 x 4 tiles of v_mfma_f32_16x16x32_bf16, 128 fp32 accumulators per lane), operands streamed HBM -> LDS by
// global_load_lds_dwordx4 (1 KiB per wave-instruction; the XOR swizzle is applied to the per-lane SOURCE chunk so the LDS image
// stays lane-linear) into 2 stages of 64 KiB, with a PHASED, STAGGERED main loop:
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
// ADDR selects the source addressing of the LDS-DMA pieces:
//   0  general: 64-bit per-lane addresses, K only needs to be a multiple of 8 (chunks past K read a zero page)
//   1  K % 64 == 0 and operand spans < 2 GiB (host-checked): the source address of a piece is a wave-uniform base (advanced by the K
//      offset on the scalar unit) + a constant 32-bit per-lane byte offset — one 64-bit add per piece instead of two adds, a compare
//      and two selects (58 -> 32 VALU instructions per K step)
//   2  ADDR 1 on SPLIT operands (MTT_SPLIT: x = hi + lo, two bf16 planes of identical layout): the fp32-class product
//      A B^T ~= Ah Bh^T + Al Bh^T + Ah Bl^T as ONE K-concatenated bf16 GEMM — K step s = 3 t + j streams the K tile t of planes
//      (A, B) = (hi, hi), (lo, hi), (hi, lo) for j = 0, 1, 2, so the loop, its hazards and its MFMA rate are exactly those of the bf16
//      kernel at 3 K; only the scalar base of a piece changes.  The re-read of a hi tile follows its first read by one / two K steps
//      (L2 hit).  Accumulation is fp32 as in the register-staged x3 kernel (gemm_kernel<.., 2>), which adds the same three products.
// (Round 2 also built a 256 x 128 tile, an implicit-GEMM conv form, a balanced DMA schedule, a persistent-workgroup form and a
// swapped-operand direct-store epilogue of this kernel; none was faster on the step's shapes — numbers in DESIGN.md section 7 — and
// they were removed from the library in round 3.)
// ---------------------------------------------------------------------------------------------
template <int ADDR>
__global__ __launch_bounds__(512, 1) void gemm_dma_kernel(const GemmP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int WAVES_N = 4, WAVES_M = 2, MT = 8, NT = 4;
  constexpr int TILE_A = BM2 * BK * 2, TILE_B = 256 * BK * 2, STAGE = TILE_A + TILE_B;
  constexpr bool FASTADDR = ADDR >= 1, X3CAT = ADDR == 2;
  MTT_TRACE_RT(0); MTT_TRACE(1);
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int tiles_n = (p.d.N + 255) / 256, tiles_m = (p.d.M + BM2 - 1) / BM2;
  int tile_m, tile_n;
  grouped_tile(wg, tiles_m, tiles_n, p.group_m, tile_m, tile_n);
  const int m0 = tile_m * BM2, n0 = tile_n * 256;
  const int z = blockIdx.z;
  const int zo = z / p.d.batch_inner, zi = z - zo * p.d.batch_inner;
  const int64_t za = (int64_t)zo * p.d.a_zo + (int64_t)zi * p.d.a_zi, zb = (int64_t)zo * p.d.b_zo + (int64_t)zi * p.d.b_zi;
  const bf16_t* Abase = (const bf16_t*)p.d.A + za;
  const bf16_t* Bbase = (const bf16_t*)p.d.B + zb;
  const bf16_t* AbaseL = X3CAT ? (const bf16_t*)p.d.A_lo + za : Abase;
  const bf16_t* BbaseL = X3CAT ? (const bf16_t*)p.d.B_lo + zb : Bbase;

  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int late = wave >> 2;                      // 1: this wave runs one phase behind (shares its SIMD with wave - 4)
  const int K = p.d.K;

  uint64_t zpage = (uint64_t)(uintptr_t)g_zero_page;
  asm volatile("" : "+s"(zpage));

  // ---- per-lane source addressing ------------------------------------------------------------------------------------
  // A: wave w streams rows [32 w, 32 w + 32) as 4 pieces of 8 rows x 128 B; lane -> (row, swizzled 16-byte chunk); B likewise
  int64_t aoff[4], boff[4];                        // element offset of this lane's (row, chunk)
  int ack[4];                                      // this lane's k offset inside a K step (chunk * 8), the same for A and B
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ lds_swz(row);
    ack[i] = c * 8;
    int ra = m0 + row; if (ra > p.d.M - 1) ra = p.d.M - 1;      // ragged edge: re-read the last valid row (results unused)
    aoff[i] = row_off((uint32_t)ra, p.d.a_mb, p.d.a_bs, p.d.lda, p.divAmb) + ack[i];
    int rb = n0 + row; if (rb > p.d.N - 1) rb = p.d.N - 1;
    boff[i] = (int64_t)rb * p.d.ldb + ack[i];
  }
  uint32_t aoff32[4], boff32[4];
  if constexpr (FASTADDR) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { aoff32[i] = (uint32_t)aoff[i] * 2u; boff32[i] = (uint32_t)boff[i] * 2u; }
  }
  // s = K step (wave-uniform).  X3CAT: K tile t = s / 3 of plane pair j = s % 3 (see the header comment)
  auto issue = [&](int stage, int s) {
    unsigned char* sA = smem + stage * STAGE + wave * 4096;
    unsigned char* sB = smem + stage * STAGE + TILE_A + wave * 4096;
    if constexpr (FASTADDR) {
      int t = s;
      const unsigned char* Ak = (const unsigned char*)Abase;
      const unsigned char* Bk = (const unsigned char*)Bbase;
      if constexpr (X3CAT) {
        t = (s * 21846) >> 16;                                  // s / 3 for s < 32768
        const int j = s - 3 * t;
        if (j == 1) Ak = (const unsigned char*)AbaseL;
        if (j == 2) Bk = (const unsigned char*)BbaseL;
      }
      Ak += (size_t)t * (BK * 2);
      Bk += (size_t)t * (BK * 2);
#pragma unroll
      for (int i = 0; i < 4; ++i) glds16((const bf16_t*)(Ak + aoff32[i]), sA + i * 1024);
#pragma unroll
      for (int i = 0; i < 4; ++i) glds16((const bf16_t*)(Bk + boff32[i]), sB + i * 1024);
    } else {
      const int k0 = s * BK;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint64_t src = (uint64_t)(uintptr_t)(Abase + (aoff[i] + k0));
        glds16((const bf16_t*)(uintptr_t)(k0 + ack[i] < K ? src : zpage), sA + i * 1024);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint64_t src = (uint64_t)(uintptr_t)(Bbase + (boff[i] + k0));
        glds16((const bf16_t*)(uintptr_t)(k0 + ack[i] < K ? src : zpage), sB + i * 1024);
      }
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = X3CAT ? 3 * (K / BK) : (K + BK - 1) / BK;
  issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                    // tile 0 is in LDS
  MTT_TRACE(2);
  if (late) __builtin_amdgcn_s_barrier();          // stagger: waves 4-7 start one slot later
  __builtin_amdgcn_sched_barrier(0);

  // the loop body is instantiated once per half, as in round 2 (one straight-line schedule per half for the register allocator).
  // (Round 3 also measured the LDS-DMA issues moved out of R0 and interleaved with the MFMAs of the C phases — early half: tile kt+1 in
  // C0(kt), late half: tile kt+2 in C1(kt) — bitwise equal and race-clean, but within +-1 % on every step shape:
  // profiles/r03_gemm_bench_f_dma_sched.log.  Where the pieces are issued is not what bounds the K loop.)
  auto main_loop = [&](auto late_tag) {
  (void)late_tag;
  for (int kt = 0; kt < nk; ++kt) {
    const unsigned char* Ah = smem + (kt & 1) * STAGE;
    const unsigned char* Bh = Ah + TILE_A;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      // ---- R phase: fragments of this 32-deep half (+ this wave's share of the next tile's LDS-DMA) ----
      if (kh == 0 && kt + 1 < nk) issue((kt + 1) & 1, kt + 1);
      u32x4 fa[MT], fb[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) fb[t] = *(const u32x4*)(Bh + lds_off(wn * 64 + t * 16 + li, kh * 4 + lg));
#pragma unroll
      for (int t = 0; t < MT; ++t) fa[t] = *(const u32x4*)(Ah + lds_off(wm * (MT * 16) + t * 16 + li, kh * 4 + lg));
      if (kh == 1) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // next tile (own part) landed
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
  }
  };
  if (late) main_loop(std::true_type{}); else main_loop(std::false_type{});
  if (!late) __builtin_amdgcn_s_barrier();         // waves 0-3 wait one slot for the late half
  __syncthreads();                                 // everyone is past its last LDS read: the epilogue may reuse the stages
  MTT_TRACE(3);
  gemm_epilogue_auto<256, WAVES_M, WAVES_N, MT, NT>(p, acc, smem, m0, n0, zo, zi);
  MTT_TRACE(4); MTT_TRACE_RT(5); MTT_TRACE_ID(tile_m * tiles_n + tile_n);
}

template <int ADDR>
int launch_dma(const GemmP& p, hipStream_t stream) {
  constexpr int smem = (BM2 * BK * 2 + 256 * BK * 2) * 2;
  static std::atomic<unsigned long long> done{0};
  if (int e = mtt_ensure_dyn_lds((const void*)gemm_dma_kernel<ADDR>, smem, done)) return e;
  const int tm = (p.d.M + BM2 - 1) / BM2, tn = (p.d.N + 255) / 256;
  dim3 grid(tm * tn, 1, p.d.batch);
  hipLaunchKernelGGL((gemm_dma_kernel<ADDR>), grid, dim3(512), smem, stream, p);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Ring layout helpers of the split-plane kernel below (round 4).  A "part" is one operand plane of a 32-deep K step: [256 rows][32 k] bf16,
// 64 B per row.  Row r lives at byte 64 r; its four 16-byte chunks at position c ^ ring_swz(r) — ds_read_b128 of a 16-row fragment
// (lane = (row li, chunk lg)) then touches 16 distinct 16-byte bank slots in each of the instruction's four lane groups
// (MI355X_MICROARCH.md, LDS table: {0-3,12-15,20-27}, {4-11,16-19,28-31} and the same + 32).  One LDS-DMA piece (a wave instruction,
// 1 KiB) is 16 rows x 64 B; the swizzle is applied to the per-lane SOURCE chunk, the LDS image stays lane-linear.
// (Round 4 also built the bf16 kernel on this ring — 32-deep slots, 4 or 5 of them, counted vmcnt, 2 S - 3 phase slots of landing time
// instead of 3 — bitwise equal to gemm_dma_kernel<1>, race-clean, and NOT faster: +-1.5 % on the step shapes, -5 % at 8192^3
// (profiles/r04_gemm_a_*.log): the K loop does not wait on memory.  And a 256 x 128 two-workgroups-per-CU kernel with software-pipelined
// waves: -10 ... -20 % (profiles/r04_gemm_d_pair.log).  Both live in the round-4 history, not in the library.)
// ---------------------------------------------------------------------------------------------
MTT_DEV int ring_swz(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }

#define MTT_VMCNT_CASE(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
// s_waitcnt vmcnt(n) for a wave-uniform run-time n (0 .. 24 even, 1 .. 9 odd; anything else waits for everything)
MTT_DEV void wait_vmcnt_dyn(int n) {
  switch (n) {
    MTT_VMCNT_CASE(24) MTT_VMCNT_CASE(22) MTT_VMCNT_CASE(20) MTT_VMCNT_CASE(18) MTT_VMCNT_CASE(16) MTT_VMCNT_CASE(14) MTT_VMCNT_CASE(12)
    MTT_VMCNT_CASE(10) MTT_VMCNT_CASE(8) MTT_VMCNT_CASE(6) MTT_VMCNT_CASE(4) MTT_VMCNT_CASE(2)
    MTT_VMCNT_CASE(9) MTT_VMCNT_CASE(7) MTT_VMCNT_CASE(5) MTT_VMCNT_CASE(3) MTT_VMCNT_CASE(1)     // the edge tile's last two steps
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

// ---------------------------------------------------------------------------------------------
// gemm_ring3_kernel (round 4): the fp32-class product on MTT_SPLIT operands as a TRUE three-product kernel — per 32-deep K step the four
// planes Ah, Al, Bh, Bl are staged ONCE (4 x 16 KiB) and feed three MFMA sets (Ah Bh^T, Ah Bl^T, Al Bh^T: 96 MFMAs per wave), where
// gemm_dma_kernel<2>'s K-concatenated form staged six plane tiles per three sets and read 36 fragments per 96 MFMAs (here: 24).
// Two slots of 64 KiB.  A step is six phases R0 C0 R1 C1 R2 C2 (R0 reads Ah + Bh, R1 reads Bl, R2 reads Al over Ah's registers), waves 4-7
// one phase behind waves 0-3 as in the bf16 kernels.  The ring is recycled PART by part: a plane's LDS is refilled right after the
// phase that read it, two steps ahead of its next use (~10 phase slots in flight instead of 3):
//     R0(j) issues Al(j+1)   [Al(j-1) was last read in R2(j-1)]      wait for Bl(j)     : late end of R0(j), early end of C0(j)
//     R1(j) issues Ah,Bh(j+2) [Ah,Bh(j) were last read in R0(j)]     wait for Al(j)     : late end of R1(j), early end of C1(j)
//     R2(j) issues Bl(j+2)   [Bl(j) was last read in R1(j)]          wait for Ah,Bh(j+1): late end of R2(j), early end of C2(j)
//   so a wave's issue order is (Ah Bh, Bl, Al) of step 0, 1, 2, ... and every wait is a counted vmcnt on the pieces issued after the
//   awaited ones (10 / 12 / 10 in steady state, less in the last two steps).  Every wait precedes the barrier that closes the time slot
//   before the first read of the awaited part by waves 0-3; every refill follows the barrier that closed the last read by waves 4-7.
// ---------------------------------------------------------------------------------------------
// CONV (MTT_OP_CONV_K on split planes: the 3x3 convs of the fp32-class forward, fea_fuse[1] / InvPT's 576-channel convs): the A operand is
// the implicit im2col of the NHWC planes, k = tap * Cp + ci with Cp % 32 == 0, so a 32-deep K step lies inside ONE tap: per step the
// scalar unit turns j into (tap, channel offset) and a signed row delta, and every lane redirects its 16-byte source to a zero page when
// that tap falls outside the image for its output pixel (a 9-bit mask per lane and piece, computed once) — two v_cndmask + one 64-bit add
// per LDS-DMA piece on top of the plain kernel.
// EDGE (round 5): an N-edge tile with at most 128 columns to compute and store (the decoder widths 300 / 350 leave 44 + 4 / 94 + 2 columns in
// their second tile; the nine-tap head GEMM 96 in its thirteenth) runs the waves as 4 x 2 instead of 2 x 4 — each wave 64 x 64 (4 x 4 MFMA tiles)
// instead of 128 x 64 — so the tile issues HALF the MFMAs; only 128 rows of the B planes are staged (one LDS-DMA piece per wave and part
// instead of two: 48 KiB per K step instead of 64).  Phases, ring recycling and hazards are those of the full tile; the counted waits
// follow from the pieces per part (nA = 2, nB = 2 | 1): steady state 3 nA + 2 nB / 3 nA + 3 nB / 2 nA + 3 nB = 10 / 12 / 10 | 8 / 9 / 7.
#ifndef MTT_R3_EDGE
#define MTT_R3_EDGE 1
#endif
template <int N_> MTT_DEV void wait_vmcnt_imm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N_) : "memory"); }

template <bool CONV, bool EDGE>
MTT_DEV void ring3_tile(const GemmP& p, unsigned char* smem, int m0, int n0, int zo, int zi) {
  constexpr int WAVES_N = EDGE ? 2 : 4, WAVES_M = EDGE ? 4 : 2, MT = EDGE ? 4 : 8, NT = 4;
  constexpr int NA = 2, NB = EDGE ? 1 : 2;                           // LDS-DMA pieces a wave issues per A / B part
  constexpr int PART = 256 * 64, SLOT = 4 * PART;                    // parts of a slot: 0 Ah, 1 Bh, 2 Bl, 3 Al
  const int64_t za = (int64_t)zo * p.d.a_zo + (int64_t)zi * p.d.a_zi, zb = (int64_t)zo * p.d.b_zo + (int64_t)zi * p.d.b_zi;
  const unsigned char* AbaseH = (const unsigned char*)((const bf16_t*)p.d.A + za);
  const unsigned char* AbaseL = (const unsigned char*)((const bf16_t*)p.d.A_lo + za);
  const unsigned char* BbaseH = (const unsigned char*)((const bf16_t*)p.d.B + zb);
  const unsigned char* BbaseL = (const unsigned char*)((const bf16_t*)p.d.B_lo + zb);

  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int late = wave >> 2;
  const int nk = p.d.K >> 5;

  uint32_t aoff32[2], boff32[2];
  unsigned tapmask[2] = {0x1ffu, 0x1ffu};                            // CONV: bit t set <=> tap t of this lane's output pixel reads inside the image
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (wave * 2 + i) * 16 + (lane >> 2);
    const int c = (lane & 3) ^ ring_swz(row);
    int ra = m0 + row; if (ra > p.d.M - 1) ra = p.d.M - 1;
    if constexpr (CONV) {
      aoff32[i] = (uint32_t)((int64_t)ra * p.d.lda + c * 8) * 2u;    // the centre pixel's row (rows = B * H * W pixels, pitch lda)
      const uint32_t q = fdiv((uint32_t)ra, p.divW);
      const int px = ra - (int)q * p.d.conv.W;
      const int py = (int)q - (int)fdiv(q, p.divH) * p.d.conv.H;
      unsigned mk = 0;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        int ty = t / 3, tx = t - 3 * (t / 3);
        if (p.d.conv.flip) { ty = 2 - ty; tx = 2 - tx; }
        const int yy = py + (ty - 1) * p.d.conv.dil, xx = px + (tx - 1) * p.d.conv.dil;
        if (yy >= 0 && yy < p.d.conv.H && xx >= 0 && xx < p.d.conv.W) mk |= 1u << t;
      }
      tapmask[i] = mk;
    } else {
      aoff32[i] = (uint32_t)(row_off((uint32_t)ra, p.d.a_mb, p.d.a_bs, p.d.lda, p.divAmb) + c * 8) * 2u;
    }
    // B rows: wave w streams rows [32 w, 32 w + 32) of the 256-row part as two pieces; EDGE: rows [16 w, 16 w + 16) of the 128 rows as one
    const int brow = EDGE ? wave * 16 + (lane >> 2) : row;
    const int cb = (lane & 3) ^ ring_swz(brow);
    int rb = n0 + brow; if (rb > p.d.N - 1) rb = p.d.N - 1;
    boff32[i] = (uint32_t)((int64_t)rb * p.d.ldb + cb * 8) * 2u;
  }
  uint64_t zpage = (uint64_t)(uintptr_t)g_zero_page;
  asm volatile("" : "+s"(zpage));
  const int cpt = CONV ? p.d.conv.Cp >> 5 : 1;                       // K steps per tap
  const int cpt_inv = ((1 << 20) + cpt - 1) / cpt;                   // j / cpt == (j * cpt_inv) >> 20 for j < 9 * cpt, cpt <= 128 (checked exhaustively)
  // this wave's pieces of part `part` (0 Ah, 1 Bh, 2 Bl, 3 Al) of K step j into slot j & 1
  auto issue_part = [&](int j, int part) {
    const bool isA = part == 0 || part == 3;
    unsigned char* dst = smem + (j & 1) * SLOT + part * PART + wave * ((EDGE && !isA) ? 1024 : 2048);
    if (CONV && isA) {
      const int tap = (j * cpt_inv) >> 20;                            // wave-uniform: scalar unit
      int ty = (tap * 11) >> 5, tx = tap - 3 * ((tap * 11) >> 5);     // tap / 3, tap % 3 for tap < 9
      if (p.d.conv.flip) { ty = 2 - ty; tx = 2 - tx; }
      const int64_t delta = ((int64_t)((ty - 1) * p.d.conv.dil * p.d.conv.W + (tx - 1) * p.d.conv.dil) * p.d.lda + (int64_t)(j - tap * cpt) * 32) * 2;
      const unsigned char* base = (part == 0 ? AbaseH : AbaseL) + delta;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const uint64_t src = (uint64_t)(uintptr_t)(base + aoff32[i]);
        glds16((const bf16_t*)(uintptr_t)(((tapmask[i] >> tap) & 1u) ? src : zpage), dst + i * 1024);
      }
      return;
    }
    const unsigned char* base = (part == 0 ? AbaseH : part == 1 ? BbaseH : part == 2 ? BbaseL : AbaseL) + (size_t)j * 64;
    glds16((const bf16_t*)(base + (isA ? aoff32[0] : boff32[0])), dst);
    if (isA || !EDGE) glds16((const bf16_t*)(base + (isA ? aoff32[1] : boff32[1])), dst + 1024);
  };
  const int fsw = ((lg ^ ring_swz(li)) << 4) + li * 64;
  const int fragA = wm * (MT * 16) * 64 + fsw, fragB = wn * (NT * 16) * 64 + fsw;

  f32x4 acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // counted waits (pieces issued AFTER the awaited part, see the header): steady state and the two last steps
  constexpr int W_BL = 3 * NA + 2 * NB, W_AL = 3 * NA + 3 * NB, W_AB = 2 * NA + 3 * NB;
  // prologue: (Ah Bh, Bl, Al)(0), (Ah Bh, Bl)(1)      [nk >= 2: the host requires K % 32 == 0 and K >= 64]
  issue_part(0, 0); issue_part(0, 1); issue_part(0, 2); issue_part(0, 3);
  issue_part(1, 0); issue_part(1, 1); issue_part(1, 2);
  wait_vmcnt_imm<2 * NA + 3 * NB>();                     // Ah, Bh of step 0 landed (this wave's pieces)
  __builtin_amdgcn_s_barrier();
  if (late) __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);

#define MTT_R3_CLOSE() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
  auto main_loop = [&](auto late_tag) {
  constexpr bool LATE = decltype(late_tag)::value;
  for (int j = 0; j < nk; ++j) {
    const unsigned char* sb = smem + (j & 1) * SLOT;
    const bool m1 = j + 1 < nk, m2 = j + 2 < nk;
    const int w_bl = NA + (m1 ? 2 * NA + 2 * NB : 0), w_al = (m1 ? 2 * NA + 2 * NB : 0) + (m2 ? NA + NB : 0), w_ab = NA + NB + (m2 ? NA + 2 * NB : 0);
    u32x4 fa[MT], fbh[NT], fbl[NT];
    // ---- R0: Ah, Bh ----
    if (m1) issue_part(j + 1, 3);
#pragma unroll
    for (int t = 0; t < NT; ++t) fbh[t] = *(const u32x4*)(sb + PART + fragB + t * 1024);
#pragma unroll
    for (int t = 0; t < MT; ++t) fa[t] = *(const u32x4*)(sb + fragA + t * 1024);
    if (LATE) { if (m2) wait_vmcnt_imm<W_BL>(); else wait_vmcnt_dyn(w_bl); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    MTT_R3_CLOSE();
    // ---- C0: Ah Bh^T ----
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int b = 0; b < NT; ++b) acc[a][b] = mfma16(fa[a], fbh[b], acc[a][b]);
    __builtin_amdgcn_s_setprio(0);
    if (!LATE) { if (m2) wait_vmcnt_imm<W_BL>(); else wait_vmcnt_dyn(w_bl); }
    MTT_R3_CLOSE();
    // ---- R1: Bl ----
    if (m2) { issue_part(j + 2, 0); issue_part(j + 2, 1); }
#pragma unroll
    for (int t = 0; t < NT; ++t) fbl[t] = *(const u32x4*)(sb + 2 * PART + fragB + t * 1024);
    if (LATE) { if (m2) wait_vmcnt_imm<W_AL>(); else wait_vmcnt_dyn(w_al); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    MTT_R3_CLOSE();
    // ---- C1: Ah Bl^T ----
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int b = 0; b < NT; ++b) acc[a][b] = mfma16(fa[a], fbl[b], acc[a][b]);
    __builtin_amdgcn_s_setprio(0);
    if (!LATE) { if (m2) wait_vmcnt_imm<W_AL>(); else wait_vmcnt_dyn(w_al); }
    MTT_R3_CLOSE();
    // ---- R2: Al (over Ah's registers) ----
    if (m2) issue_part(j + 2, 2);
#pragma unroll
    for (int t = 0; t < MT; ++t) fa[t] = *(const u32x4*)(sb + 3 * PART + fragA + t * 1024);
    if (LATE) { if (m2) wait_vmcnt_imm<W_AB>(); else wait_vmcnt_dyn(w_ab); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    MTT_R3_CLOSE();
    // ---- C2: Al Bh^T ----
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int b = 0; b < NT; ++b) acc[a][b] = mfma16(fa[a], fbh[b], acc[a][b]);
    __builtin_amdgcn_s_setprio(0);
    if (!LATE) { if (m2) wait_vmcnt_imm<W_AB>(); else wait_vmcnt_dyn(w_ab); }
    MTT_R3_CLOSE();
  }
  };
#undef MTT_R3_CLOSE
  if (late) main_loop(std::true_type{}); else main_loop(std::false_type{});
  if (!late) __builtin_amdgcn_s_barrier();
  __syncthreads();
  if constexpr (EDGE) gemm_epilogue<128, WAVES_M, WAVES_N, MT, NT>(p, acc, smem, m0, n0, zo, zi);      // ragged by construction: the general epilogue
  else gemm_epilogue_auto<256, WAVES_M, WAVES_N, MT, NT>(p, acc, smem, m0, n0, zo, zi);
}

template <bool CONV>
__global__ __launch_bounds__(512, 1) void gemm_ring3_kernel(const GemmP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int tiles_n = (p.d.N + 255) / 256, tiles_m = (p.d.M + BM2 - 1) / BM2;
  int tile_m, tile_n;
  grouped_tile(wg, tiles_m, tiles_n, p.group_m, tile_m, tile_n);
  const int m0 = tile_m * BM2, n0 = tile_n * 256;
  const int z = blockIdx.z;
  const int zo = z / p.d.batch_inner, zi = z - zo * p.d.batch_inner;
  // workgroup-uniform: the columns this tile computes or zero-fills (n_store: the channel padding of the output pitch)
  const int ncols = (p.d.n_store > p.d.N ? p.d.n_store : p.d.N) - n0;
  if (MTT_R3_EDGE && ncols <= 128 && p.d.store_mode == MTT_STORE_ROWS) ring3_tile<CONV, true>(p, smem, m0, n0, zo, zi);
  else ring3_tile<CONV, false>(p, smem, m0, n0, zo, zi);
}

template <bool CONV>
int launch_ring3(const GemmP& p, hipStream_t stream) {
  constexpr int smem = 2 * 4 * 16384;
  static std::atomic<unsigned long long> done{0};
  if (int e = mtt_ensure_dyn_lds((const void*)gemm_ring3_kernel<CONV>, smem, done)) return e;
  const int tm = (p.d.M + BM2 - 1) / BM2, tn = (p.d.N + 255) / 256;
  dim3 grid(tm * tn, 1, p.d.batch);
  hipLaunchKernelGGL((gemm_ring3_kernel<CONV>), grid, dim3(512), smem, stream, p);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// gemm_ringc_kernel (round 5): the implicit-GEMM 3x3 conv in bf16 on the LDS-DMA tile — gemm_dma_kernel<1>'s tile, wave layout, staggered
// R | C phases and epilogue with the operands in a RING of S = 4 slots of 32-deep K sub-tiles (the round-4 ring of the bf16 kernel,
// measured equal to gemm_dma_kernel<1> on plain GEMMs; what it adds here is the 32-deep step that lies inside ONE tap whenever the channel
// pitch is a multiple of 32) and gemm_ring3_kernel<true>'s per-step tap addressing: the scalar unit turns the step index into (tap, channel
// offset) and a signed row delta, every lane redirects its 16-byte source to a zero page when that tap leaves the image for its output
// pixel.  Takes the bf16 3x3 convs that gemm_kernel<MTT_OP_CONV_K, ., 0> (register-staged, 640-720 TFLOP/s) ran: the input gradients of
// every 3x3 conv in the bf16 / x3f backward (mirrored taps on the transposed pack), and the forward convs of the bf16 mode.
// A slot = [A 256 x 32 | B 256 x 32] bf16 = 32 KiB; sub-tile j + 3 is issued in R(j) into the slot sub-tile j - 1 left; waits are counted:
// after the 4 pieces of sub-tile j + 1 a wave has issued those of j + 2 and j + 3 (8) in steady state.
// ---------------------------------------------------------------------------------------------
// EDGE: an N-edge tile of <= 128 columns (350 = 256 + 94 + 2 of padding) runs the waves 4 x 2 with 64 x 64 wave tiles — half the MFMAs — and
// stages 128 B rows (one piece per wave: 3 pieces per sub-tile instead of 4), as gemm_ring3_kernel does.
template <bool EDGE>
MTT_DEV void ringc_tile(const GemmP& p, unsigned char* smem, int m0, int n0, int zo, int zi) {
  constexpr int WAVES_N = EDGE ? 2 : 4, WAVES_M = EDGE ? 4 : 2, MT = EDGE ? 4 : 8, NT = 4, S = 4;
  constexpr int NP = EDGE ? 3 : 4;                                   // LDS-DMA pieces a wave issues per sub-tile
  constexpr int PART = 256 * 64, SLOT = 2 * PART;                    // [256 rows][32 k] bf16 per operand
  const unsigned char* Abase = (const unsigned char*)((const bf16_t*)p.d.A + ((int64_t)zo * p.d.a_zo + (int64_t)zi * p.d.a_zi));
  const unsigned char* Bbase = (const unsigned char*)((const bf16_t*)p.d.B + ((int64_t)zo * p.d.b_zo + (int64_t)zi * p.d.b_zi));

  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int late = wave >> 2;
  const int nk = p.d.K >> 5;                                          // 32-deep sub-tiles: K = 9 * Cp, Cp % 32 == 0

  uint32_t aoff32[2], boff32[2];
  unsigned tapmask[2];                                                // bit t set <=> tap t of this lane's output pixel reads inside the image
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (wave * 2 + i) * 16 + (lane >> 2);
    const int c = (lane & 3) ^ ring_swz(row);
    int ra = m0 + row; if (ra > p.d.M - 1) ra = p.d.M - 1;
    aoff32[i] = (uint32_t)((int64_t)ra * p.d.lda + c * 8) * 2u;      // the centre pixel's row (rows = B * H * W pixels, pitch lda)
    const uint32_t q = fdiv((uint32_t)ra, p.divW);
    const int px = ra - (int)q * p.d.conv.W;
    const int py = (int)q - (int)fdiv(q, p.divH) * p.d.conv.H;
    unsigned mk = 0;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      int ty = t / 3, tx = t - 3 * (t / 3);
      if (p.d.conv.flip) { ty = 2 - ty; tx = 2 - tx; }
      const int yy = py + (ty - 1) * p.d.conv.dil, xx = px + (tx - 1) * p.d.conv.dil;
      if (yy >= 0 && yy < p.d.conv.H && xx >= 0 && xx < p.d.conv.W) mk |= 1u << t;
    }
    tapmask[i] = mk;
    const int brow = EDGE ? wave * 16 + (lane >> 2) : row;            // EDGE: rows [16 w, 16 w + 16) of the 128 B rows, one piece
    const int cb = (lane & 3) ^ ring_swz(brow);
    int rb = n0 + brow; if (rb > p.d.N - 1) rb = p.d.N - 1;
    boff32[i] = (uint32_t)((int64_t)rb * p.d.ldb + cb * 8) * 2u;
  }
  uint64_t zpage = (uint64_t)(uintptr_t)g_zero_page;
  asm volatile("" : "+s"(zpage));
  const int cpt = p.d.conv.Cp >> 5;                                   // K steps per tap
  const int cpt_inv = ((1 << 20) + cpt - 1) / cpt;                    // j / cpt == (j * cpt_inv) >> 20 for j < 9 * cpt, cpt <= 128
  // the four pieces (A rows 2 w, 2 w + 1 of 16; B likewise) of sub-tile j into ring slot `slot`
  auto issue_tile = [&](int j, int slot) {
    unsigned char* dst = smem + slot * SLOT + wave * 2048;
    const int tap = (j * cpt_inv) >> 20;                              // wave-uniform: scalar unit
    int ty = (tap * 11) >> 5, tx = tap - 3 * ((tap * 11) >> 5);       // tap / 3, tap % 3 for tap < 9
    if (p.d.conv.flip) { ty = 2 - ty; tx = 2 - tx; }
    const int64_t delta = ((int64_t)((ty - 1) * p.d.conv.dil * p.d.conv.W + (tx - 1) * p.d.conv.dil) * p.d.lda + (int64_t)(j - tap * cpt) * 32) * 2;
    const unsigned char* abase = Abase + delta;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const uint64_t src = (uint64_t)(uintptr_t)(abase + aoff32[i]);
      glds16((const bf16_t*)(uintptr_t)(((tapmask[i] >> tap) & 1u) ? src : zpage), dst + i * 1024);
    }
    const unsigned char* bbase = Bbase + (size_t)j * 64;
    if constexpr (EDGE) {
      glds16((const bf16_t*)(bbase + boff32[0]), smem + slot * SLOT + PART + wave * 1024);
    } else {
      glds16((const bf16_t*)(bbase + boff32[0]), dst + PART);
      glds16((const bf16_t*)(bbase + boff32[1]), dst + PART + 1024);
    }
  };
  const int fsw = ((lg ^ ring_swz(li)) << 4) + li * 64;
  const int fragA = wm * (MT * 16) * 64 + fsw, fragB = PART + wn * (NT * 16) * 64 + fsw;

  f32x4 acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int npro = nk < S - 1 ? nk : S - 1;                           // nk >= 9
  for (int j = 0; j < npro; ++j) issue_tile(j, j);
  wait_vmcnt_dyn(NP * (npro - 1));
  __builtin_amdgcn_s_barrier();                    // sub-tile 0 is in LDS
  if (late) __builtin_amdgcn_s_barrier();          // stagger: waves 4-7 start one slot later
  __builtin_amdgcn_sched_barrier(0);

  auto main_loop = [&](auto late_tag) {
  constexpr bool LATE = decltype(late_tag)::value;
  int rslot = 0, wslot = (S - 1) % S;
  for (int j = 0; j < nk; ++j) {
    const unsigned char* sb = smem + rslot * SLOT;
    const bool more = j + S - 1 < nk;              // this step issues sub-tile j + S - 1
    // the wait of this step covers sub-tile j + 1: behind its pieces the wave has issued sub-tiles j + 2 .. min(j + S - 1, nk - 1)
    const int newer = more ? S - 2 : (nk - j - 2 > 0 ? nk - j - 2 : 0);
    // ---- R phase ----
    if (more) issue_tile(j + S - 1, wslot);
    u32x4 fa[MT], fb[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) fb[t] = *(const u32x4*)(sb + fragB + t * 1024);
#pragma unroll
    for (int t = 0; t < MT; ++t) fa[t] = *(const u32x4*)(sb + fragA + t * 1024);
    if (LATE) { if (more) wait_vmcnt_imm<2 * NP>(); else wait_vmcnt_dyn(NP * newer); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
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
    if (!LATE) { if (more) wait_vmcnt_imm<2 * NP>(); else wait_vmcnt_dyn(NP * newer); }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    rslot = rslot + 1 == S ? 0 : rslot + 1;
    wslot = wslot + 1 == S ? 0 : wslot + 1;
  }
  };
  if (late) main_loop(std::true_type{}); else main_loop(std::false_type{});
  if (!late) __builtin_amdgcn_s_barrier();         // waves 0-3 wait one slot for the late half
  __syncthreads();                                 // everyone is past its last LDS read: the epilogue may reuse the ring
  if constexpr (EDGE) gemm_epilogue<128, WAVES_M, WAVES_N, MT, NT>(p, acc, smem, m0, n0, zo, zi);
  else gemm_epilogue_auto<256, WAVES_M, WAVES_N, MT, NT>(p, acc, smem, m0, n0, zo, zi);
}

__global__ __launch_bounds__(512, 1) void gemm_ringc_kernel(const GemmP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int tiles_n = (p.d.N + 255) / 256, tiles_m = (p.d.M + BM2 - 1) / BM2;
  int tile_m, tile_n;
  grouped_tile(wg, tiles_m, tiles_n, p.group_m, tile_m, tile_n);
  const int m0 = tile_m * BM2, n0 = tile_n * 256;
  const int z = blockIdx.z;
  const int zo = z / p.d.batch_inner, zi = z - zo * p.d.batch_inner;
  const int ncols = (p.d.n_store > p.d.N ? p.d.n_store : p.d.N) - n0;
  if (MTT_R3_EDGE && ncols <= 128) ringc_tile<true>(p, smem, m0, n0, zo, zi);
  else ringc_tile<false>(p, smem, m0, n0, zo, zi);
}

int launch_ringc(const GemmP& p, hipStream_t stream) {
  constexpr int smem = 4 * 32768;
  static std::atomic<unsigned long long> done{0};
  if (int e = mtt_ensure_dyn_lds((const void*)gemm_ringc_kernel, smem, done)) return e;
  const int tm = (p.d.M + BM2 - 1) / BM2, tn = (p.d.N + 255) / 256;
  dim3 grid(tm * tn, 1, p.d.batch);
  hipLaunchKernelGGL(gemm_ringc_kernel, grid, dim3(512), smem, stream, p);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// gemm_f32n_kernel (round 4): D[M, N <= 32] = A[M, K] W[N, K]^T + bias in EXACT fp32 on the matrix pipe (v_mfma_f32_32x32x2_f32: f32 in,
// f32 accumulate, one rounding per product — bitwise a k-ordered fmaf chain, cdna_hip_programming.md §3).  For the fp32-class modes' tall
// GEMMs with a handful of outputs (the heads' 1x1 predictions on the full-resolution maps: M = B 128 128 rows, K = 352, N = 1 .. 21): they
// are HBM-bound on reading A once, and the split-product kernels spend three bf16 MFMA products on a 128-wide tile of which 21 columns are
// real.  The f32 MFMA runs at 1 / 16 of the bf16 rate, but 32 columns at that rate are still 2x faster than A can be streamed.
//   One wave = 32 rows; lane (r = l & 31, h = l >> 5) loads 16 bytes A[row r][8 i + 4 h .. + 3] per step i and feeds four MFMAs (the
//   reduction order inside a step is permuted consistently on both operands: instruction (i, e) contracts k = 8 i + 4 h + e).  W sits in LDS as
//   [K / 4][32][4] (chunk-major: the 32 lanes of a half read 512 contiguous bytes), rows >= N zero.  Workgroup = 4 waves x RB row blocks.
// ---------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(16))) float f32x16;
constexpr int F32N_RB = 4;            // 32-row blocks per wave
// the A prologue of one 4-channel chunk: a <- act(a * scale + shift), optionally stored rounded to bf16 (8 bytes per lane)
MTT_DEV void f32n_prologue(float4& a, const float4 sc, const float4 sh, int act, bf16_t* a16, int64_t off16) {
  a.x = fmaf(a.x, sc.x, sh.x); a.y = fmaf(a.y, sc.y, sh.y); a.z = fmaf(a.z, sc.z, sh.z); a.w = fmaf(a.w, sc.w, sh.w);
  if (act == MTT_ACT_GELU) { a.x = gelu_f(a.x); a.y = gelu_f(a.y); a.z = gelu_f(a.z); a.w = gelu_f(a.w); }
  else if (act == MTT_ACT_RELU) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }
  if (a16) *(u32x2*)(a16 + off16) = (u32x2){pack2(a.x, a.y), pack2(a.z, a.w)};
}
__global__ __launch_bounds__(256, 2) void gemm_f32n_kernel(const GemmP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float4* wl = (float4*)smem;                                        // [K / 4][32] chunks of 4 k
  const mtt_gemm_desc& d = p.d;
  const int z = blockIdx.z;
  const int zo = z / d.batch_inner, zi = z - zo * d.batch_inner;
  const float* A = (const float*)d.A + ((int64_t)zo * d.a_zo + (int64_t)zi * d.a_zi);
  const float* W = (const float*)d.B + ((int64_t)zo * d.b_zo + (int64_t)zi * d.b_zi);
  float* D = (float*)d.D + ((int64_t)zo * d.d_zo + (int64_t)zi * d.d_zi);
  const int K4 = d.K >> 2;
  for (int t = threadIdx.x; t < K4 * 32; t += 256) {
    const int n = t & 31, q = t >> 5;
    wl[t] = n < d.N ? *(const float4*)(W + (int64_t)n * d.ldb + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // A prologue (mtt_gemm_desc.a_scale): per-channel scale / shift chunks behind the weights, [K / 4] float4 each
  const bool pro = d.a_scale != nullptr;
  float4* const scl = wl + K4 * 32;
  float4* const shl = scl + K4;
  if (pro)
    for (int t = threadIdx.x; t < K4; t += 256) { scl[t] = *(const float4*)(d.a_scale + 4 * t); shl[t] = *(const float4*)(d.a_shift + 4 * t); }
  bf16_t* const a16 = (bf16_t*)d.a_aux16;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int n_store = d.n_store > d.N ? d.n_store : d.N;
  const float bias = (d.colshift && r < d.N) ? d.colshift[(int64_t)zo * d.col_zo + (int64_t)zi * d.col_zi + r] : 0.f;
  const int steps = d.K >> 3;
#pragma unroll 1
  for (int rb = 0; rb < F32N_RB; ++rb) {
    const int64_t m0 = ((int64_t)blockIdx.x * 4 * F32N_RB + (int64_t)rb * 4 + wave) * 32;
    if (m0 >= d.M) break;
    int64_t row = m0 + r; if (row > d.M - 1) row = d.M - 1;
    const float* ar = A + row * d.lda + 4 * h;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    int i = 0;
    for (; i + 4 <= steps; i += 4) {                               // four 16-byte loads in flight per lane
      float4 a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) a[u] = *(const float4*)(ar + 8 * (i + u));
#pragma unroll
      for (int u = 0; u < 4; ++u) b[u] = wl[(2 * (i + u) + h) * 32 + r];
      if (pro) {
#pragma unroll
        for (int u = 0; u < 4; ++u) f32n_prologue(a[u], scl[2 * (i + u) + h], shl[2 * (i + u) + h], d.a_act, a16, row * d.ld_a16 + 8 * (i + u) + 4 * h);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].x, b[u].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].y, b[u].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].z, b[u].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].w, b[u].w, acc, 0, 0, 0);
      }
    }
    for (; i < steps; ++i) {
      float4 a = *(const float4*)(ar + 8 * i);
      const float4 b = wl[(2 * i + h) * 32 + r];
      if (pro) f32n_prologue(a, scl[2 * i + h], shl[2 * i + h], d.a_act, a16, row * d.ld_a16 + 8 * i + 4 * h);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
    }
    // acc[g] = D[row (g & 3) + 8 (g >> 2) + 4 h][col r]: lanes of a half write 32 consecutive columns of one row
    if (r < n_store) {
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const int64_t m = m0 + (g & 3) + 8 * (g >> 2) + 4 * h;
        if (m < d.M) D[m * d.ldd + r] = r < d.N ? acc[g] + bias : 0.f;
      }
    }
  }
}

static bool f32n_ok(const mtt_gemm_desc& d) {
  return d.prec == MTT_PREC_X3 && d.a_dtype == MTT_F32 && d.b_dtype == MTT_F32 && d.d_dtype == MTT_F32 && d.a_op == MTT_OP_K && d.b_op == MTT_OP_K &&
         d.N <= 32 && d.K % 8 == 0 && d.K <= 1024 && (d.M >= 2048 || d.a_scale) && d.store_mode == MTT_STORE_ROWS && d.a_mb <= 0 && d.d_mb <= 0 && !d.resid &&
         !d.aux_in && !d.aux_out && !d.colscale && !d.rowscale && !d.colsum_out && d.act == MTT_ACT_NONE && d.alpha == 1.0f &&
         (d.variant == MTT_GEMM_AUTO) && d.lda % 4 == 0 && d.ldb % 4 == 0 &&
         (!d.a_scale || (d.a_shift && d.batch <= 1 && (d.a_act == MTT_ACT_NONE || d.a_act == MTT_ACT_GELU || d.a_act == MTT_ACT_RELU) &&
                         (!d.a_aux16 || (d.ld_a16 % 4 == 0 && d.ld_a16 >= d.K && !((uintptr_t)d.a_aux16 & 7))) &&
                         !(((uintptr_t)d.a_scale | (uintptr_t)d.a_shift) & 15)));
}
int launch_f32n(const GemmP& p, hipStream_t stream) {
  const int smem = p.d.K * 32 * 4 + 2 * p.d.K * 4;                  // weights + the prologue's scale / shift vectors
  static std::atomic<unsigned long long> done{0};
  if (int e = mtt_ensure_dyn_lds((const void*)gemm_f32n_kernel, 1024 * 32 * 4 + 2 * 1024 * 4, done)) return e;
  const int64_t rows_per_wg = 4 * F32N_RB * 32;
  dim3 grid((unsigned)((p.d.M + rows_per_wg - 1) / rows_per_wg), 1, p.d.batch);
  hipLaunchKernelGGL(gemm_f32n_kernel, grid, dim3(256), smem, stream, p);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// gemm_dma128_kernel<FASTADDR, NT>: 128 x (32 NT) x 64 tile, 4 waves (2 x 2, each 64 rows x 16 NT columns).  NT = 4: 128 x 128, 64 fp32 accumulators per lane,
// operands streamed by LDS-DMA into 2 stages of 32 KiB — TWO workgroups per CU, so a SIMD alternates between a wave of each: while
// one waits for its K tile the other issues MFMAs; NT = 1 / 2: 32 / 64-column tiles for outputs of a few channels (head predictions:
// HBM-bound on reading A once, 40 KiB of LDS, four workgroups per CU).  For the K-contiguous bf16 GEMMs whose N is too narrow for the 256-wide tile
// (decoder 1x1 convs with 300 / 350 output channels, head predictions): the register-staged general kernel runs them latency-bound
// (~25 % MFMA issue at K = 1024: global -> VGPR -> LDS with one K step of lookahead), and a 256-wide tile pair wastes 41 / 32 % of its
// columns.  One K step: issue the LDS-DMA of tile kt + 1, read the fragments of tile kt, 32 MFMAs, wait for the DMA, barrier.
//   RAW  tile kt + 1 is read after the barrier that follows every wave's vmcnt(0).
//   WAR  stage (kt + 1) & 1 held tile kt - 1, whose last fragment reads completed (lgkmcnt(0)) before the barrier closing step kt - 1.
// (Also measured in round 3, then removed: a 256 x 128 x 32 form of this loop — the MAIN kernel's 128 x 64 wave tile, three 24-KiB stages with
// counted vmcnt, two workgroups per CU so that one's prologue / epilogue runs under the other's K loop.  Bitwise equal and race-clean, but
// 1 050 vs 1 411 TFLOP/s at 8192^3 and 745 vs 873 on the qkv shape: one barrier per 32 MFMAs costs more than the overlap returns —
// profiles/r03_gemm_bench_o_v5.log.)
// ---------------------------------------------------------------------------------------------
template <bool FASTADDR, int NT>
__global__ __launch_bounds__(256, 2) void gemm_dma128_kernel(const GemmP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int MT = 4, TBN = 32 * NT, TILE_A = 128 * BK * 2, TILE_B = TBN * BK * 2, STAGE = TILE_A + TILE_B;
  constexpr int BP = TBN >= 128 ? 4 : (TBN / 8 < 4 ? TBN / 8 : 4);     // B pieces (8 rows each) per streaming wave; waves past the tile stream none
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int tiles_n = (p.d.N + TBN - 1) / TBN, tiles_m = (p.d.M + 127) / 128;
  int tile_m, tile_n;
  grouped_tile(wg, tiles_m, tiles_n, 2 * p.group_m, tile_m, tile_n);
  const int m0 = tile_m * 128, n0 = tile_n * TBN;
  const int z = blockIdx.z;
  const int zo = z / p.d.batch_inner, zi = z - zo * p.d.batch_inner;
  const bf16_t* Abase = (const bf16_t*)p.d.A + ((int64_t)zo * p.d.a_zo + (int64_t)zi * p.d.a_zi);
  const bf16_t* Bbase = (const bf16_t*)p.d.B + ((int64_t)zo * p.d.b_zo + (int64_t)zi * p.d.b_zi);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  const int K = p.d.K;
  uint64_t zpage = (uint64_t)(uintptr_t)g_zero_page;
  asm volatile("" : "+s"(zpage));

  // wave w streams rows [32 w, 32 w + 32) of both tiles as 4 pieces of 8 rows x 128 B; lane -> (row, swizzled 16-byte chunk)
  int64_t aoff[4], boff[4];
  int ack[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ lds_swz(row);
    ack[i] = c * 8;
    int ra = m0 + row; if (ra > p.d.M - 1) ra = p.d.M - 1;      // ragged edge: re-read the last valid row (results unused)
    aoff[i] = row_off((uint32_t)ra, p.d.a_mb, p.d.a_bs, p.d.lda, p.divAmb) + ack[i];
    int rb = n0 + row; if (rb > p.d.N - 1) rb = p.d.N - 1;
    boff[i] = (int64_t)rb * p.d.ldb + ack[i];
  }
  uint32_t aoff32[4], boff32[4];
  if constexpr (FASTADDR) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { aoff32[i] = (uint32_t)aoff[i] * 2u; boff32[i] = (uint32_t)boff[i] * 2u; }
  }
  auto issue = [&](int stage, int s) {
    unsigned char* sA = smem + stage * STAGE + wave * 4096;
    unsigned char* sB = smem + stage * STAGE + TILE_A + wave * 4096;
    const bool bw = wave * 32 < TBN;                             // this wave streams B rows [32 w, 32 w + 32) if the tile has them
    if constexpr (FASTADDR) {
      const unsigned char* Ak = (const unsigned char*)Abase + (size_t)s * (BK * 2);
      const unsigned char* Bk = (const unsigned char*)Bbase + (size_t)s * (BK * 2);
#pragma unroll
      for (int i = 0; i < 4; ++i) glds16((const bf16_t*)(Ak + aoff32[i]), sA + i * 1024);
      if (bw) {
#pragma unroll
        for (int i = 0; i < BP; ++i) glds16((const bf16_t*)(Bk + boff32[i]), sB + i * 1024);
      }
    } else {
      const int k0 = s * BK;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint64_t src = (uint64_t)(uintptr_t)(Abase + (aoff[i] + k0));
        glds16((const bf16_t*)(uintptr_t)(k0 + ack[i] < K ? src : zpage), sA + i * 1024);
      }
      if (bw) {
#pragma unroll
        for (int i = 0; i < BP; ++i) {
          const uint64_t src = (uint64_t)(uintptr_t)(Bbase + (boff[i] + k0));
          glds16((const bf16_t*)(uintptr_t)(k0 + ack[i] < K ? src : zpage), sB + i * 1024);
        }
      }
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = (K + BK - 1) / BK;
  issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int kt = 0; kt < nk; ++kt) {
    const unsigned char* Ah = smem + (kt & 1) * STAGE;
    const unsigned char* Bh = Ah + TILE_A;
    if (kt + 1 < nk) issue((kt + 1) & 1, kt + 1);
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      u32x4 fa[MT], fb[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) fb[t] = *(const u32x4*)(Bh + lds_off(wn * (NT * 16) + t * 16 + li, kh * 4 + lg));
#pragma unroll
      for (int t = 0; t < MT; ++t) fa[t] = *(const u32x4*)(Ah + lds_off(wm * 64 + t * 16 + li, kh * 4 + lg));
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = mfma16(fa[a], fb[b], acc[a][b]);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // own share of tile kt + 1 landed; own reads of tile kt done
    __builtin_amdgcn_s_barrier();
  }
  gemm_epilogue_auto<TBN, 2, 2, MT, NT>(p, acc, smem, m0, n0, zo, zi);
}

template <bool FASTADDR, int NT>
int launch_dma128(const GemmP& p, hipStream_t stream) {
  constexpr int TBN = 32 * NT;
  constexpr int stage = (128 + TBN) * BK * 2, ep = 64 * (TBN + 4) * 4 + 256 * 8 * 4;   // K-loop stages / epilogue staging + column-sum slots
  constexpr int smem = 2 * stage > ep ? 2 * stage : ep;     // 64 KiB at TBN = 128 (two workgroups per CU), 40 KiB at TBN = 32 (four)
  static std::atomic<unsigned long long> done{0};
  if (int e = mtt_ensure_dyn_lds((const void*)gemm_dma128_kernel<FASTADDR, NT>, smem, done)) return e;
  const int tm = (p.d.M + 127) / 128, tn = (p.d.N + TBN - 1) / TBN;
  dim3 grid(tm * tn, 1, p.d.batch);
  hipLaunchKernelGGL((gemm_dma128_kernel<FASTADDR, NT>), grid, dim3(256), smem, stream, p);
  return (int)hipGetLastError();
}
// column tile 32 / 64 / 128: narrow outputs get a narrow tile; so do SMALL problems (the channel attention's prompt-row GEMMs: 378 x 1024 x 1024
// is 24 tiles of 128 x 128 on 256 CUs) — they are bound by the latency of their K loop, not by MFMA issue, and narrower tiles put it on more CUs
static int dma128_nt(const mtt_gemm_desc& d) {
  if (d.N <= 32) return 1;
  if (d.N <= 64) return 2;
  const int batch = d.batch < 1 ? 1 : d.batch;
  const int64_t tm = (d.M + 127) / 128;
  if (tm * ((d.N + 127) / 128) * batch >= 64) return 4;
  return tm * ((d.N + 63) / 64) * batch >= 64 ? 2 : 1;
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

// MEDGE (round 5): an M-edge tile with at most 128 output rows (the decoder layers' 300 / 350 outputs leave 44 / 94 rows in their second row
// tile, InvPT's 576 leave 64) runs the waves as 1 x 8 — each wave all 128 rows x 32 columns (8 x 2 MFMA tiles) instead of 128 x 64 — so the
// tile issues half the MFMAs and half the B fragment reads; the staging is unchanged (chunks of A beyond M read the zero page as before).
template <bool CONVB, bool MEDGE>
MTT_DEV void tn_tile(const GemmP& p, unsigned char* smem, int m0, int n0, int zo, int zi) {
  constexpr int MT = 8, NT = MEDGE ? 2 : 4, WAVES_N = MEDGE ? 8 : 4, WAVES_M = MEDGE ? 1 : 2;
  constexpr int TILE = BK * 256 * 2, STAGE = 2 * TILE;          // 32 KiB per operand tile, 64 KiB per stage
  const bf16_t* Abase = (const bf16_t*)p.d.A + ((int64_t)zo * p.d.a_zo + (int64_t)zi * p.d.a_zi);
  const bf16_t* Bbase = (const bf16_t*)p.d.B + ((int64_t)zo * p.d.b_zo + (int64_t)zi * p.d.b_zi);

  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = MEDGE ? 0 : wave >> 2, wn = MEDGE ? wave : wave & 3;
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
    const int unit = wn * NT + t;
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
#define TNW_A TNW(al[0]), TNW(al[1]), TNW(al[2]), TNW(al[3]), TNW(al[4]), TNW(al[5]), TNW(al[6]), TNW(al[7]), \
              TNW(ah[0]), TNW(ah[1]), TNW(ah[2]), TNW(ah[3]), TNW(ah[4]), TNW(ah[5]), TNW(ah[6]), TNW(ah[7])
      if constexpr (MEDGE) {
        if (kh == 1) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : TNW(bl[0]), TNW(bl[1]), TNW(bh[0]), TNW(bh[1]), TNW_A :: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" : TNW(bl[0]), TNW(bl[1]), TNW(bh[0]), TNW(bh[1]), TNW_A :: "memory");
      } else {
        if (kh == 1)
          asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
                       : TNW(bl[0]), TNW(bl[1]), TNW(bl[2]), TNW(bl[3]), TNW(bh[0]), TNW(bh[1]), TNW(bh[2]), TNW(bh[3]), TNW_A :: "memory");
        else
          asm volatile("s_waitcnt lgkmcnt(0)"
                       : TNW(bl[0]), TNW(bl[1]), TNW(bl[2]), TNW(bl[3]), TNW(bh[0]), TNW(bh[1]), TNW(bh[2]), TNW(bh[3]), TNW_A :: "memory");
      }
#undef TNW_A
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
  if constexpr (MEDGE) gemm_epilogue<256, WAVES_M, WAVES_N, MT, NT>(p, acc, smem, m0, n0, zo, zi);      // 128 x 256 tile, ragged rows: general epilogue
  else gemm_epilogue_auto<256, WAVES_M, WAVES_N, MT, NT>(p, acc, smem, m0, n0, zo, zi);
}

template <bool CONVB>
__global__ __launch_bounds__(512, 1) void gemm_tn_kernel(const GemmP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int tiles_n = (p.d.N + 255) / 256, tiles_m = (p.d.M + 255) / 256;
  int tile_m, tile_n;
  grouped_tile(wg, tiles_m, tiles_n, p.group_m, tile_m, tile_n);
  const int m0 = tile_m * 256, n0 = tile_n * 256;
  const int z = blockIdx.z;
  const int zo = z / p.d.batch_inner, zi = z - zo * p.d.batch_inner;
  if (MTT_R3_EDGE && p.d.M - m0 <= 128 && !p.d.colsum_out && p.d.store_mode == MTT_STORE_ROWS) tn_tile<CONVB, true>(p, smem, m0, n0, zo, zi);
  else tn_tile<CONVB, false>(p, smem, m0, n0, zo, zi);
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
    case 22: return sizeof(mtt_segcopy_desc);
    case 23: return sizeof(mtt_ctrw_desc);
    case 24: return sizeof(mtt_detloss_desc);
    default: return 0;
  }
}

// Kernel choice, a pure function of the descriptor.  Return codes (also what mtt_gemm_variant reports):
//   0 register-staged 128 x 128 (general: any operand layout / dtype / precision)
//   3 LDS-DMA 256 x 256 phased / staggered (gemm_dma_kernel)
//   4 LDS-DMA 128 x 128, two workgroups per CU (gemm_dma128_kernel)
//   6 token-major weight-gradient kernel (gemm_tn_kernel): LDS-DMA + ds_read_b64_tr_b16 fragments
//   8 gemm_ring3_kernel<false>: MTT_SPLIT operands, fp32-class product (three MFMA products per staged K step)
//   9 gemm_ring3_kernel<true>: the same with the implicit im2col of a 3x3 conv as A operand (MTT_OP_CONV_K on planes)
//  11 gemm_f32n_kernel: fp32 operands, N <= 32, exact fp32 MFMA (the fp32-class modes' tall few-output GEMMs)
//  12 gemm_ringc_kernel: bf16 implicit-GEMM 3x3 conv on the LDS-DMA ring (channel pitch % 32 == 0)
//  <0 MTT_E_* (no kernel takes this descriptor)
// d.variant = MTT_GEMM_AUTO applies the policy; MTT_GEMM_GENERAL / MTT_GEMM_DMA256 force a kernel where it is applicable.
// K % 64 == 0 and 32-bit per-lane byte offsets from the batch member's base: the fast source addressing of the LDS-DMA kernel
static bool dma_fastaddr_ok(const mtt_gemm_desc& d, int kmod = 64) {
  if (d.K % kmod) return false;
  const int64_t a_span = (d.a_mb > 0 ? (int64_t)((d.M - 1) / d.a_mb) * d.a_bs + (int64_t)((d.M - 1) % d.a_mb) * d.lda : (int64_t)(d.M - 1) * d.lda) + d.K;
  const int64_t b_span = (int64_t)(d.N - 1) * d.ldb + d.K;
  return a_span < (1ll << 31) && b_span < (1ll << 31);
}
static int gemm_variant_for(const mtt_gemm_desc& d) {
  const bool any_split = d.a_dtype == MTT_SPLIT || d.b_dtype == MTT_SPLIT;
  if (any_split) {
    // pre-split planes exist for ONE kernel: both operands split, reduction-contiguous, whole 32-deep K steps (at least two), fast addressing
    const bool both = d.prec == MTT_PREC_X3 && d.a_dtype == MTT_SPLIT && d.b_dtype == MTT_SPLIT && d.b_op == MTT_OP_K && d.A_lo && d.B_lo &&
                      d.store_mode == MTT_STORE_ROWS;
    if (d.a_scale) return MTT_E_UNSUPPORTED;
    if (both && d.a_op == MTT_OP_K && dma_fastaddr_ok(d, MTT_RING ? 32 : 64) && d.K >= 64 && d.K / 64 * 3 < 32768) return 8;
    // ... and its implicit-GEMM form for the 3x3 convs: channel pitch a multiple of 32 (a K step inside one tap), pixel rows of pitch lda
    if (both && d.a_op == MTT_OP_CONV_K && d.conv.Cp % 32 == 0 && d.conv.Cp <= 4096 && d.K == 9 * d.conv.Cp && d.a_mb <= 0 &&
        (int64_t)d.M * d.lda < (1ll << 31) && (int64_t)(d.N - 1) * d.ldb + d.K < (1ll << 31)) return 9;
    return MTT_E_UNSUPPORTED;
  }
  if (f32n_ok(d)) return 11;
  if (d.a_scale) return MTT_E_UNSUPPORTED;             // the A prologue exists in that kernel only
  if (d.prec != MTT_PREC_BF16 || d.a_dtype != MTT_BF16 || d.b_dtype != MTT_BF16) return 0;
  // 12: the bf16 implicit-GEMM 3x3 conv on the LDS-DMA ring (gemm_ringc_kernel): channel pitch a multiple of 32, pixel rows of pitch lda,
  // 32-bit element offsets; enough rows and columns for 256 x 256 tiles (small maps stay on the register-staged 128 x 128 kernel)
  if (d.a_op == MTT_OP_CONV_K && d.b_op == MTT_OP_K && d.variant != MTT_GEMM_GENERAL && d.store_mode == MTT_STORE_ROWS && d.conv.Cp % 32 == 0 &&
      d.conv.Cp <= 4096 && d.K == 9 * d.conv.Cp && d.lda == d.conv.Cp && d.a_mb <= 0 && (int64_t)d.M * d.lda < (1ll << 31) &&
      (int64_t)(d.N - 1) * d.ldb + d.K < (1ll << 31) && !d.colsum_out &&
      (d.variant == MTT_GEMM_DMA256 || (d.M >= 2048 && d.N >= 128))) return 12;
  // 6: token-major weight-gradient kernel (gemm_tn_kernel): both operands MTT_OP_R (B may be the implicit im2col^T), bf16
  const bool tn = d.a_op == MTT_OP_R && (d.b_op == MTT_OP_R || d.b_op == MTT_OP_CONV_R) && d.store_mode == MTT_STORE_ROWS;
  if (tn && d.variant != MTT_GEMM_GENERAL) {
    if (d.variant == MTT_GEMM_DMA256) return 6;
    const int64_t pm = (d.M + 255) / 256 * 256, pn = (d.N + 255) / 256 * 256;
    const int batch = d.batch < 1 ? 1 : d.batch;
    const bool fills = (pm / 256) * (pn / 256) * batch >= 64 && d.K >= 512;
    // 256-wide tiles may waste up to ~55 % of the MFMA work and still beat the 128-wide general kernel on token-major operands (decoder
    // outputs of 300 / 350 channels: 432 vs 697 us, 313 vs 424, 232 vs 286 at B = 63, profiles/r03_dec_wgrad_bench_h.log)
    if (fills && 100 * (int64_t)d.M * d.N >= 45 * pm * pn) return 6;
    return 0;
  }
  const bool dma = d.a_op == MTT_OP_K && d.b_op == MTT_OP_K && (d.K % 8) == 0;
  if (!dma || d.variant == MTT_GEMM_GENERAL) return 0;
  if (d.variant == MTT_GEMM_DMA256) return 3;
  if (d.variant == MTT_GEMM_DMA128) return 4;
  // AUTO.  Measured on MI355X (profiles/r02_gemm_bench_b*.log, r02_conv_bench_b.log, B = 63 shapes): the 256 x 256 DMA tile wins for
  // wide outputs (qkv / proj / fc1 / fc2: 830-1170 vs 610-740 TFLOP/s on the register-staged 128 x 128 kernel); narrow decoder shapes
  // (N = 300 / 350: 22 % of a 256-wide tile pair is padding) and few-tile problems stay on the general kernel.
  const int n256 = (d.N + 255) / 256 * 256, n128 = (d.N + 127) / 128 * 128;
  const bool wide = 100 * n128 > 85 * n256;              // a 128-wide tiling would save < 15 % of the columns
  const int batch = d.batch < 1 ? 1 : d.batch;
  const int64_t blocks = (int64_t)((d.M + 255) / 256) * ((d.N + 255) / 256) * batch;
  if (wide && d.M >= 512 && d.N >= 512 && blocks >= 96) return 3;
  // 4: the 128 x 128 LDS-DMA kernel (two workgroups per CU) for what is left: narrow outputs (decoder 1x1s with 300 / 350 channels, head
  // predictions) and mid-size problems — whenever there is enough work to fill the chip's 512 workgroup slots
  const int tbn = 32 * dma128_nt(d);                      // narrow outputs (head predictions: 1 - 21 classes) get a 32 / 64-column tile
  const int64_t blocks128 = (int64_t)((d.M + 127) / 128) * ((d.N + tbn - 1) / tbn) * batch;
  if (d.M >= 512 && d.K >= 128 && blocks128 >= 256) return 4;
  // small problems with a long reduction (few tiles, >= 8 K steps): the LDS-DMA loop has a third of the register-staged loop's K-step latency
  if (d.M >= 128 && d.N >= 256 && d.K >= 512) return 4;
  return 0;
}
#ifdef MTT_GEMM_TRACE
static unsigned long long* g_trace_host = nullptr;
extern "C" void mtt_debug_set_trace(void* buf) { g_trace_host = (unsigned long long*)buf; }
#endif
extern "C" int mtt_gemm_variant(const mtt_gemm_desc* d) { return d ? gemm_variant_for(*d) : MTT_E_BADARG; }

static int gemm_launch(GemmP& p, hipStream_t s, int v);
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
  if ((d.aux_in || d.aux_out) && (d.ldaux <= 0 || (d.ldaux % 8))) return MTT_E_BADARG;
  if (d.act < MTT_ACT_NONE || d.act > MTT_ACT_MUL_AUX) return MTT_E_BADARG;
  if ((d.act == MTT_ACT_GELU_DAUX && !d.aux_out) || ((d.act == MTT_ACT_MUL_AUX || d.act == MTT_ACT_GELU_BWD || d.act == MTT_ACT_RELU_BWD) && !d.aux_in))
    return MTT_E_BADARG;
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
  p.group_m = MTT_GROUP_M;
#ifdef MTT_GEMM_TRACE
  p.trace = g_trace_host;
#endif
  hipStream_t s = (hipStream_t)stream;
  const int v = gemm_variant_for(d);
  if (v < 0) return v;
  if (d.d_dtype == MTT_SPLIT && !d.D_lo) return MTT_E_BADARG;
  if (d.d_dtype == MTT_SPLIT && d.store_mode != MTT_STORE_ROWS) return MTT_E_UNSUPPORTED;
  if (d.colsum_out) {                                   // bias gradient from the epilogue: per-row-block partials + a fixed-order second stage
    if (!d.colsum_ws) return MTT_E_BADARG;
    if (d.batch != 1 || d.store_mode != MTT_STORE_ROWS || v == 6) return MTT_E_UNSUPPORTED;
    const int rc = gemm_launch(p, s, v);
    if (rc) return rc;
    const int tbm = (v == 3 || v == 8 || v == 9) ? 256 : BM;           // (the 128 x 128 LDS-DMA kernel has the general kernel's row block)
    hipLaunchKernelGGL(mtt_colsum_final_kernel, dim3((d.N + 31) / 32, 1, 1), dim3(256), 0, s, (const float*)d.colsum_ws, d.colsum_out, d.N,
                       (d.M + tbm - 1) / tbm, (int64_t)0);
    return (int)hipGetLastError();
  }
  return gemm_launch(p, s, v);
}

extern "C" size_t mtt_gemm_colsum_ws_floats(const mtt_gemm_desc* d) {
  if (!d || d->M <= 0 || d->N <= 0) return 0;
  return (size_t)((d->M + BM - 1) / BM) * (size_t)((d->N + 7) / 8 * 8);     // the smallest row block of any kernel variant
}

static int gemm_launch(GemmP& p, hipStream_t s, int v) {
  mtt_gemm_desc& d = p.d;
  if (v == 11) return launch_f32n(p, s);
  if (v == 12) return launch_ringc(p, s);
  if (v == 9) return launch_ring3<true>(p, s);
#if MTT_RING
  if (v == 8) return launch_ring3<false>(p, s);
#else
  if (v == 8) return launch_dma<2>(p, s);
#endif
  if (v == 3) return dma_fastaddr_ok(d) ? launch_dma<1>(p, s) : launch_dma<0>(p, s);
  if (v == 4) {
    const bool fa = dma_fastaddr_ok(d);
    switch (dma128_nt(d)) {
      case 1: return fa ? launch_dma128<true, 1>(p, s) : launch_dma128<false, 1>(p, s);
      case 2: return fa ? launch_dma128<true, 2>(p, s) : launch_dma128<false, 2>(p, s);
      default: return fa ? launch_dma128<true, 4>(p, s) : launch_dma128<false, 4>(p, s);
    }
  }
  if (v == 6) return d.b_op == MTT_OP_CONV_R ? launch_tn<true>(p, s) : launch_tn<false>(p, s);
  // general kernel.  MODE: 0 bf16 x bf16; 1 A f32 (rounded while staged) x bf16; 2 x3 (both f32, split while staged);
  // 3 f32 x f32, 4 bf16 x f32, rounded while staged (bf16 arithmetic on fp32-stored tensors: the backward of the x3-forward training mode)
  int mode;
  if (d.prec == MTT_PREC_X3) {
    if (d.a_dtype != MTT_F32 || d.b_dtype != MTT_F32) return MTT_E_UNSUPPORTED;
    mode = 2;
  } else if (d.a_dtype == MTT_BF16 && d.b_dtype == MTT_BF16) mode = 0;
  else if (d.a_dtype == MTT_F32 && d.b_dtype == MTT_BF16) mode = 1;
  else if (d.a_dtype == MTT_F32 && d.b_dtype == MTT_F32) { mode = 3; if (d.colsum_out) return MTT_E_UNSUPPORTED; }
  else if (d.a_dtype == MTT_BF16 && d.b_dtype == MTT_F32) mode = 4;
  else return MTT_E_UNSUPPORTED;
#define MTT_CASE(AO, BO) \
  if (d.a_op == AO && d.b_op == BO) \
    return mode == 2 ? launch<AO, BO, 2>(p, s) : (mode == 1 ? launch<AO, BO, 1>(p, s) : (mode == 3 ? launch<AO, BO, 3>(p, s) : (mode == 4 ? launch<AO, BO, 4>(p, s) : launch<AO, BO, 0>(p, s))));
  MTT_CASE(MTT_OP_K, MTT_OP_K)
  MTT_CASE(MTT_OP_K, MTT_OP_R)
  MTT_CASE(MTT_OP_R, MTT_OP_R)
  MTT_CASE(MTT_OP_CONV_K, MTT_OP_K)
  MTT_CASE(MTT_OP_R, MTT_OP_CONV_R)
#undef MTT_CASE
  return MTT_E_UNSUPPORTED;
}
