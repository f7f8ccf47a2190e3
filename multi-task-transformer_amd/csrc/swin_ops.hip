// TaskPrompter-Swin forward kernels (TaskPrompter/models/transformers/taskprompter_swin.py): general patchify, row gather (window
// partition / reverse / patch merging), window attention with the task prompts joined to every window (bf16: MFMA, swapped product
// so that softmax statistics are per-lane; fp32: exact VALU kernel for the parity mode), channel attention of the prompts over the
// feature channels, and the small-channel 3x3 stride-2 convolution of the attention maps.
#include "mtt_device.h"

namespace {

// ------------------------------------------------------------------------------------------------
// Patchify, k = s = P: one thread per (patch, c, dy) moves P consecutive pixels.
// ------------------------------------------------------------------------------------------------
template <int P>
__global__ __launch_bounds__(256) void patchify_p_kernel(const float* img, void* cols, int B, int H, int W, int64_t ldc, int out_dtype) {
  const int gh = H / P, gw = W / P;
  const int KP = 3 * P * P;
  const int per_patch = 3 * P + 1;                 // 3*P row pieces + one thread that zeroes the padding columns
  const int64_t total = (int64_t)B * gh * gw * per_patch;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int piece = (int)(t % per_patch);
    const int64_t patch = t / per_patch;
    if (piece == 3 * P) {
      for (int k = KP; k < ldc; ++k) st_elem(cols, patch * ldc + k, out_dtype, 0.0f);
      continue;
    }
    const int c = piece / P, dy = piece % P;
    const int px = (int)(patch % gw), py = (int)((patch / gw) % gh), b = (int)(patch / ((int64_t)gw * gh));
    const float* src = img + (((int64_t)b * 3 + c) * H + py * P + dy) * W + px * P;
#pragma unroll
    for (int dx = 0; dx < P; ++dx) st_elem(cols, patch * ldc + (c * P + dy) * P + dx, out_dtype, src[dx]);
  }
}

// ------------------------------------------------------------------------------------------------
// Row gather with zero fill: one thread per 8 columns of a destination row.
// ------------------------------------------------------------------------------------------------
MTT_DEV void ld8g(const void* p, int64_t idx, int dtype, float (&v)[8]) {
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
MTT_DEV void st8g(void* p, int64_t idx, int dtype, const float (&v)[8]) {
  if (dtype == MTT_BF16) {
    *(u32x4*)((bf16_t*)p + idx) = (u32x4){pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
  } else {
    *(float4*)((float*)p + idx) = make_float4(v[0], v[1], v[2], v[3]);
    *(float4*)((float*)p + idx + 4) = make_float4(v[4], v[5], v[6], v[7]);
  }
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const mtt_gather_desc d) {
  const int chunks = d.C / 8;
  const int64_t total = (int64_t)d.B * d.rows * chunks;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int ch = (int)(t % chunks);
    const int64_t r = (t / chunks) % d.rows;
    const int64_t b = t / ((int64_t)chunks * d.rows);
    const int src_row = d.idx[b * d.idx_bs + r];
    if (src_row < 0 && d.skip_neg) continue;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (src_row >= 0) ld8g(d.src, b * d.src_bs + (int64_t)src_row * d.ld_src + ch * 8, d.src_dtype, v);
    st8g(d.dst, b * d.dst_bs + r * d.ld_dst + ch * 8, d.dst_dtype, v);
  }
}

// ------------------------------------------------------------------------------------------------
// Window attention, bf16 storage: one workgroup (4 waves) per (window, head), head_dim 32 = ONE 16x16x32 MFMA per 16 x 16 score tile.
//   S^T tile j = K_j Q_i^T  ->  s[j][r] = S[query 16 i + li][key 16 j + 4 lg + r]: a lane owns one query, so the softmax statistics
//   are per-lane scalars (+ 2 xor-shuffles over the 4 lane groups);
//   the C layout of two P^T tiles is a valid B fragment of O^T += V^T P^T once the A side (V^T, transposed through LDS once per
//   workgroup) reads its keys in the same permuted order (two 8-byte reads): P never touches LDS.
// Q / K fragments are 16-byte global loads (rows are 64 contiguous bytes; K is re-read from L2 by the 10 query tiles).
// NKT = number of 16-key tiles (even): N <= 16 NKT.
// ------------------------------------------------------------------------------------------------
struct __attribute__((packed, aligned(4))) F4u { float x, y, z, w; };     // 16-byte load that only promises 4-byte alignment

template <int NKT>
__global__ __launch_bounds__(256, 2) void winattn_bf16_kernel(const mtt_winattn_desc d) {
  constexpr int NP = NKT * 16, PITCH = NP + 8;     // V^T rows: NP keys (+ 8 bf16 of padding: 2-way instead of 8-way bank conflicts)
  __shared__ __attribute__((aligned(16))) bf16_t vT[32 * PITCH];
  const int nH = d.nH, T = d.T, ws2 = d.ws2, N = T + ws2, C = nH * 32, ld = 3 * C;
  const int win = blockIdx.x / nH, h = blockIdx.x % nH;
  const int wl = win % d.nW, b = win / d.nW;
  const bf16_t* base = (const bf16_t*)d.qkv + (int64_t)win * N * ld + h * 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;

  // ---- V^T into LDS (zeros for keys >= N: P is 0 there, but 0 * garbage must stay finite) ----
  for (int key = tid; key < NP; key += 256) {
    u32x4 v4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v4[q] = key < N ? *(const u32x4*)(base + (2 * C + key * ld + q * 8)) : (u32x4){0u, 0u, 0u, 0u};
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        vT[(q * 8 + 2 * e) * PITCH + key] = (bf16_t)(v4[q][e] & 0xffffu);
        vT[(q * 8 + 2 * e + 1) * PITCH + key] = (bf16_t)(v4[q][e] >> 16);
      }
  }
  __syncthreads();

  const float* bias_h = d.bias + (int64_t)h * ws2 * ws2;
  const float* mask_w = d.mask ? d.mask + (int64_t)wl * ws2 * ws2 : nullptr;
  const int nqt = (N + 15) / 16;
  for (int qt = wave; qt < nqt; qt += 4) {
    const int query = qt * 16 + li;
    const u32x4 zero4 = (u32x4){0u, 0u, 0u, 0u};
    const u32x4 qf = query < N ? *(const u32x4*)(base + (query * ld + lg * 8)) : zero4;
    f32x4 s[NKT];
#pragma unroll
    for (int j = 0; j < NKT; ++j) {
      const int krow = j * 16 + li;
      const u32x4 kf = krow < N ? *(const u32x4*)(base + (C + krow * ld + lg * 8)) : zero4;
      s[j] = mfma16(kf, qf, (f32x4){0.f, 0.f, 0.f, 0.f});
    }
    // raw prompt logits -> image layout
    if (qt == 0 && li < T && d.rawmap) {
      float* row = d.rawmap + (((int64_t)b * nH + h) * T + li) * d.map_ld + d.map_off;
      const int32_t* px = d.pix + (int64_t)wl * ws2;
#pragma unroll
      for (int j = 0; j < NKT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = j * 16 + lg * 4 + r;
          if (key >= T && key < N) {
            const int pp = px[key - T];
            if (pp >= 0) row[pp] = s[j][r];
          }
        }
    }
    // scale, bias, mask, softmax over the keys of this lane's query.  A lane's 4 keys of a tile are consecutive, so bias / mask come as
    // one (4-byte aligned) 16-byte load per tile wherever the 4 keys are all window positions; scalar loads at the edges.
    const bool qwin = query >= T && query < N;
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < NKT; ++j) {
      const int kb = j * 16 + lg * 4;
      float add[4] = {0.f, 0.f, 0.f, 0.f};
      if (qwin && kb >= T && kb + 3 < N) {
        const int o = (query - T) * ws2 + (kb - T);               // 32-bit offsets from uniform bases
        const F4u bb = *(const F4u*)(bias_h + o);
        add[0] = bb.x; add[1] = bb.y; add[2] = bb.z; add[3] = bb.w;
        if (mask_w) { const F4u mm = *(const F4u*)(mask_w + o); add[0] += mm.x; add[1] += mm.y; add[2] += mm.z; add[3] += mm.w; }
      } else if (qwin) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = kb + r;
          if (key >= T && key < N) {
            const int o = (query - T) * ws2 + (key - T);
            add[r] = bias_h[o] + (mask_w ? mask_w[o] : 0.f);
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = s[j][r] * d.scale + add[r];
        if (kb + r >= N) v = -INFINITY;
        s[j][r] = v;
        mx = fmaxf(mx, v);
      }
    }
    mx = groups_max(mx);
    float l = 0.f;
#pragma unroll
    for (int j = 0; j < NKT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pv = __expf(s[j][r] - mx);
        s[j][r] = pv;
        l += pv;
      }
    l = groups_sum(l);
    // O^T += V^T P^T over 32-key chunks
    f32x4 o[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int c = 0; c < NKT / 2; ++c) {
      const u32x4 pb = (u32x4){pack2(s[2 * c][0], s[2 * c][1]), pack2(s[2 * c][2], s[2 * c][3]),
                               pack2(s[2 * c + 1][0], s[2 * c + 1][1]), pack2(s[2 * c + 1][2], s[2 * c + 1][3])};
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const bf16_t* vr = vT + (dt * 16 + li) * PITCH + 32 * c + 4 * lg;
        const u32x2 lo = *(const u32x2*)vr;
        const u32x2 hi = *(const u32x2*)(vr + 16);
        o[dt] = mfma16((u32x4){lo[0], lo[1], hi[0], hi[1]}, pb, o[dt]);
      }
    }
    if (query < N) {
      const float inv = 1.0f / l;
      bf16_t* op = (bf16_t*)d.out + ((int64_t)win * N + query) * C + h * 32 + lg * 4;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
        *(u32x2*)(op + dt * 16) = (u32x2){pack2(o[dt][0] * inv, o[dt][1] * inv), pack2(o[dt][2] * inv, o[dt][3] * inv)};
    }
  }
}

// fp32 storage (the x3 parity mode): exact VALU arithmetic, one thread per query, K / V staged in LDS; two passes over the keys
// (max, then exp / sum / PV) so that no N-long score row is kept.
__global__ __launch_bounds__(256) void winattn_f32_kernel(const mtt_winattn_desc d) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* ks = (float*)smem_raw;                    // [N][33]
  const int nH = d.nH, T = d.T, ws2 = d.ws2, N = T + ws2, C = nH * 32, ld = 3 * C;
  float* vs = ks + N * 33;                         // [N][33]
  const int win = blockIdx.x / nH, h = blockIdx.x % nH;
  const int wl = win % d.nW, b = win / d.nW;
  const float* base = (const float*)d.qkv + (int64_t)win * N * ld + h * 32;
  const int tid = threadIdx.x;
  for (int e = tid; e < N * 32; e += 256) {
    const int key = e >> 5, dd = e & 31;
    ks[key * 33 + dd] = base[C + (int64_t)key * ld + dd];
    vs[key * 33 + dd] = base[2 * C + (int64_t)key * ld + dd];
  }
  __syncthreads();
  const int query = tid;
  if (query >= N) return;
  float q[32];
#pragma unroll
  for (int dd = 0; dd < 32; ++dd) q[dd] = base[(int64_t)query * ld + dd];
  const float* bias_h = d.bias + (int64_t)h * ws2 * ws2;
  const float* mask_w = d.mask ? d.mask + (int64_t)wl * ws2 * ws2 : nullptr;
  const bool qwin = query >= T;
  float* rawrow = (query < T && d.rawmap) ? d.rawmap + (((int64_t)b * nH + h) * T + query) * d.map_ld + d.map_off : nullptr;
  const int32_t* px = d.pix + (int64_t)wl * ws2;
  auto logit = [&](int key, float& raw) {
    float a = 0.f;
#pragma unroll
    for (int dd = 0; dd < 32; ++dd) a = fmaf(q[dd], ks[key * 33 + dd], a);
    raw = a;
    float v = a * d.scale;
    if (qwin && key >= T) {
      const int64_t o = (int64_t)(query - T) * ws2 + (key - T);
      v += bias_h[o];
      if (mask_w) v += mask_w[o];
    }
    return v;
  };
  float mx = -INFINITY;
  for (int key = 0; key < N; ++key) {
    float raw;
    const float v = logit(key, raw);
    mx = fmaxf(mx, v);
    if (rawrow && key >= T) {
      const int pp = px[key - T];
      if (pp >= 0) rawrow[pp] = raw;
    }
  }
  float l = 0.f, o[32];
#pragma unroll
  for (int dd = 0; dd < 32; ++dd) o[dd] = 0.f;
  for (int key = 0; key < N; ++key) {
    float raw;
    const float pv = expf(logit(key, raw) - mx);
    l += pv;
#pragma unroll
    for (int dd = 0; dd < 32; ++dd) o[dd] = fmaf(pv, vs[key * 33 + dd], o[dd]);
  }
  const float inv = 1.0f / l;
  float* op = (float*)d.out + ((int64_t)win * N + query) * C + h * 32;
#pragma unroll
  for (int dd = 0; dd < 32; ++dd) op[dd] = o[dd] * inv;
}

// ------------------------------------------------------------------------------------------------
// Window attention backward (first version: exact fp32 VALU arithmetic for both storage dtypes; one workgroup per (window, head), the
// five [N][32] operands Q, K, V, dO, O staged in LDS).  With P = softmax(S), D_i = dO_i . O_i:
//   dS_ij = P_ij (dO_i . v_j - D_i);  g_ij = scale dS_ij + draw_ij   (draw = gradient of the raw prompt-row logits, image layout)
//   dq_i = sum_j g_ij k_j   (phase 1, a thread per query: row max / sum first)      dk_j = sum_i g_ij q_i,  dv_j = sum_i P_ij dO_i
//   (phase 2, a thread per key);  dS of the window x window part goes to dS_out [nwin, nH, ws2, ws2] when given (the relative-position
//   bias gradient is its sum over windows, gathered by the host through relative_position_index).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void winattn_bwd_kernel(const mtt_winattn_desc d, const void* dout, const float* drawmap, void* dqkv, float* dS_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int nH = d.nH, T = d.T, ws2 = d.ws2, N = T + ws2, C = nH * 32, ld = 3 * C;
  float* qs = (float*)smem_raw;                    // [N][33] each
  float* ks = qs + N * 33;
  float* vs = ks + N * 33;
  float* gs = vs + N * 33;                         // dO
  float* os = gs + N * 33;                         // O
  float* mrow = os + N * 33;                       // [N] row max, [N] 1 / row sum, [N] D
  float* linv = mrow + N;
  float* Drow = linv + N;
  const int win = blockIdx.x / nH, h = blockIdx.x % nH;
  const int wl = win % d.nW, b = win / d.nW;
  const int64_t base = (int64_t)win * N * ld + h * 32, obase = (int64_t)win * N * C + h * 32;
  const int tid = threadIdx.x;
  for (int e = tid; e < N * 32; e += 256) {
    const int row = e >> 5, dd = e & 31;
    qs[row * 33 + dd] = ld_elem(d.qkv, base + (int64_t)row * ld + dd, d.dtype);
    ks[row * 33 + dd] = ld_elem(d.qkv, base + C + (int64_t)row * ld + dd, d.dtype);
    vs[row * 33 + dd] = ld_elem(d.qkv, base + 2 * C + (int64_t)row * ld + dd, d.dtype);
    gs[row * 33 + dd] = ld_elem(dout, obase + (int64_t)row * C + dd, d.dtype);
    os[row * 33 + dd] = ld_elem(d.out, obase + (int64_t)row * C + dd, d.dtype);
  }
  __syncthreads();
  const float* bias_h = d.bias + (int64_t)h * ws2 * ws2;
  const float* mask_w = d.mask ? d.mask + (int64_t)wl * ws2 * ws2 : nullptr;
  const int32_t* px = d.pix + (int64_t)wl * ws2;
  auto logit = [&](int i, int j) {
    float a = 0.f;
#pragma unroll
    for (int dd = 0; dd < 32; ++dd) a = fmaf(qs[i * 33 + dd], ks[j * 33 + dd], a);
    float v = a * d.scale;
    if (i >= T && j >= T) {
      const int64_t o = (int64_t)(i - T) * ws2 + (j - T);
      v += bias_h[o];
      if (mask_w) v += mask_w[o];
    }
    return v;
  };
  auto draw = [&](int i, int j) {                  // gradient of the raw logit (i < T prompt row, j >= T window pixel), else 0
    if (!drawmap || i >= T || j < T) return 0.f;
    const int pp = px[j - T];
    return pp >= 0 ? drawmap[(((int64_t)b * nH + h) * T + i) * d.map_ld + d.map_off + pp] : 0.f;
  };
  // ---- phase 1: a thread per query ----
  if (tid < N) {
    const int i = tid;
    float mx = -INFINITY;
    for (int j = 0; j < N; ++j) mx = fmaxf(mx, logit(i, j));
    float l = 0.f;
    for (int j = 0; j < N; ++j) l += expf(logit(i, j) - mx);
    float D = 0.f;
#pragma unroll
    for (int dd = 0; dd < 32; ++dd) D = fmaf(gs[i * 33 + dd], os[i * 33 + dd], D);
    mrow[i] = mx; linv[i] = 1.0f / l; Drow[i] = D;
    float dq[32];
#pragma unroll
    for (int dd = 0; dd < 32; ++dd) dq[dd] = 0.f;
    for (int j = 0; j < N; ++j) {
      const float pv = expf(logit(i, j) - mx) / l;
      float dp = 0.f;
#pragma unroll
      for (int dd = 0; dd < 32; ++dd) dp = fmaf(gs[i * 33 + dd], vs[j * 33 + dd], dp);
      const float g = d.scale * pv * (dp - D) + draw(i, j);
#pragma unroll
      for (int dd = 0; dd < 32; ++dd) dq[dd] = fmaf(g, ks[j * 33 + dd], dq[dd]);
    }
#pragma unroll
    for (int dd = 0; dd < 32; ++dd) st_elem(dqkv, base + (int64_t)i * ld + dd, d.dtype, dq[dd]);
  }
  __syncthreads();
  // ---- phase 2: a thread per key ----
  if (tid < N) {
    const int j = tid;
    float dk[32], dv[32];
#pragma unroll
    for (int dd = 0; dd < 32; ++dd) { dk[dd] = 0.f; dv[dd] = 0.f; }
    for (int i = 0; i < N; ++i) {
      const float pv = expf(logit(i, j) - mrow[i]) * linv[i];
      float dp = 0.f;
#pragma unroll
      for (int dd = 0; dd < 32; ++dd) dp = fmaf(gs[i * 33 + dd], vs[j * 33 + dd], dp);
      const float ds = pv * (dp - Drow[i]);
      const float g = d.scale * ds + draw(i, j);
#pragma unroll
      for (int dd = 0; dd < 32; ++dd) { dk[dd] = fmaf(g, qs[i * 33 + dd], dk[dd]); dv[dd] = fmaf(pv, gs[i * 33 + dd], dv[dd]); }
      if (dS_out && i >= T && j >= T) dS_out[(((int64_t)win * nH + h) * ws2 + (i - T)) * ws2 + (j - T)] = ds;
    }
#pragma unroll
    for (int dd = 0; dd < 32; ++dd) {
      st_elem(dqkv, base + C + (int64_t)j * ld + dd, d.dtype, dk[dd]);
      st_elem(dqkv, base + 2 * C + (int64_t)j * ld + dd, d.dtype, dv[dd]);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Channel attention in two kernels (the first version ran everything of a (b, t, window) in ONE workgroup: 8 workgroups on the chip,
// 267 us per call = 10 % of the Swin-B forward, profiles/r02_prof_swin_q_forward_b4_first.txt):
//   logits: grid (C / 256, B*T*nwin) — a thread owns one feature channel c (coalesced kT rows), loops over the window's elements;
//   mix   : grid (P / 32,  B*T*nwin) — every workgroup recomputes the softmax statistics over the C logits (a few KB from L2), then
//           its 4 waves take 8 window elements each: lanes stride over c, wave reduction.
// ------------------------------------------------------------------------------------------------
struct ChanWin { int b, t, win, r, wh, ww, wy, wx, P; };
MTT_DEV ChanWin chan_win(const mtt_chanattn_desc& d, int id) {
  ChanWin w;
  const int nwin = d.nh * d.nw;
  w.win = id % nwin; w.t = (id / nwin) % d.T; w.b = id / (nwin * d.T);
  w.r = (int)(sqrtf((float)d.ce) + 0.5f);
  w.wh = w.r / d.nh; w.ww = w.r / d.nw; w.P = w.wh * w.ww;
  w.wy = w.win / d.nw; w.wx = w.win % d.nw;
  return w;
}
MTT_DEV int chan_elem(const ChanWin& w, int e) { return (w.wy * w.wh + e / w.ww) * w.r + w.wx * w.ww + e % w.ww; }

__global__ __launch_bounds__(256) void chanattn_logits_kernel(const mtt_chanattn_desc d) {
  __shared__ float qs[1024], bk[1024];
  __shared__ int es[1024];
  const ChanWin w = chan_win(d, blockIdx.y);
  const int tid = threadIdx.x;
  for (int e = tid; e < w.P; e += 256) {
    const int el = chan_elem(w, e);
    es[e] = el;
    qs[e] = d.q[((int64_t)w.b * d.T + w.t) * d.ce + el];
    bk[e] = d.kvbias ? d.kvbias[el] : 0.f;
  }
  __syncthreads();
  const int c = blockIdx.x * 256 + tid;
  if (c >= d.C) return;
  const int64_t kvb = (int64_t)w.b * 2 * d.ce * d.ldk;
  float a = 0.f;
  for (int e = 0; e < w.P; ++e) a = fmaf(qs[e], ld_elem(d.kvT, kvb + (int64_t)es[e] * d.ldk + c, d.kv_dtype) + bk[e], a);
  d.rawchan[(((int64_t)w.b * d.T + w.t) * (d.nh * d.nw) + w.win) * d.C + c] = a;
}

__global__ __launch_bounds__(256) void chanattn_mix_kernel(const mtt_chanattn_desc d) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* pr = (float*)smem_raw;                    // [C] probabilities (unnormalised)
  __shared__ float red[8];
  const ChanWin w = chan_win(d, blockIdx.y);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* raw = d.rawchan + (((int64_t)w.b * d.T + w.t) * (d.nh * d.nw) + w.win) * d.C;
  float mx = -INFINITY;
  for (int c = tid; c < d.C; c += 256) { const float v = raw[c] * d.scale; pr[c] = v; mx = fmaxf(mx, v); }
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float l = 0.f;
  for (int c = tid; c < d.C; c += 256) { const float pv = expf(pr[c] - mx); pr[c] = pv; l += pv; }
  l = wave_sum(l);
  if (lane == 0) red[4 + wave] = l;
  __syncthreads();
  const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
  const int64_t kvb = (int64_t)w.b * 2 * d.ce * d.ldk;
  for (int k = 0; k < 8; ++k) {
    const int e = blockIdx.x * 32 + wave * 8 + k;
    if (e >= w.P) break;
    const int el = chan_elem(w, e);
    const int64_t vrow = kvb + (int64_t)(d.ce + el) * d.ldk;
    float a = 0.f;
    for (int c = lane; c < d.C; c += 64) a = fmaf(pr[c], ld_elem(d.kvT, vrow + c, d.kv_dtype), a);
    a = wave_sum(a);
    if (lane == 0) d.cx[((int64_t)w.b * d.T + w.t) * d.ce + el] = a * inv + (d.kvbias ? d.kvbias[d.ce + el] : 0.f);     // sum_c p_c = 1
  }
}

// ------------------------------------------------------------------------------------------------
// 3x3 stride-2 pad-1 convolution on small-channel fp32 maps: one thread per output element.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv3s2_kernel(const mtt_conv3s2_desc d) {
  const int Ho = d.H / 2, Wo = d.W / 2;
  const int64_t total = (int64_t)d.B * d.Co * Ho * Wo;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int xo = (int)(t % Wo), yo = (int)((t / Wo) % Ho), co = (int)((t / ((int64_t)Wo * Ho)) % d.Co);
    const int64_t b = t / ((int64_t)Wo * Ho * d.Co);
    float a = d.bias ? d.bias[co] : 0.f;
    for (int ci = 0; ci < d.Ci; ++ci) {
      const float* xp = d.x + b * d.x_bs + (int64_t)ci * d.x_cs + d.x_off;
      const float* wp = d.w + ((int64_t)co * d.Ci + ci) * 9;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int yy = 2 * yo - 1 + ky;
        if (yy < 0 || yy >= d.H) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int xx = 2 * xo - 1 + kx;
          if (xx < 0 || xx >= d.W) continue;
          a = fmaf(xp[(int64_t)yy * d.W + xx], wp[ky * 3 + kx], a);
        }
      }
    }
    d.y[b * d.y_bs + (int64_t)co * d.y_cs + d.y_off + (int64_t)yo * Wo + xo] = a;
  }
}


// ------------------------------------------------------------------------------------------------
// Backward of the channel attention (autograd of taskprompter_swin.py:399-409), three small kernels over the forward's layouts:
//   A  per (b, t, window): P = softmax_c(scale * rawchan), dP[c] = sum_{e in win} dcx[e] * vT[e, c],
//      dS[c] = scale * P[c] * (dP[c] - sum_c' P dP) + drawchan[c]       -> workspace planes dS, P  [B*T*nwin, C]
//   B  dq[b, t, e]   = sum_c dS[c] * kT[e, c]                           (one wave per window element)
//   C  dkT[b, e, c]  = sum_t q[b, t, e] * dS[b, t, win(e), c];   dvT[b, e, c] = sum_t dcx[b, t, e] * P[b, t, win(e), c]
// Fixed summation orders throughout (no atomics).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void chanattn_bwd_a_kernel(const mtt_chanattn_desc d, const float* drawchan, const float* dcx, float* ws) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* pr = (float*)smem_raw;                    // [C] probabilities
  __shared__ float red[8];
  __shared__ float gs[1024];
  __shared__ int es[1024];
  const ChanWin w = chan_win(d, blockIdx.x);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t grp = ((int64_t)w.b * d.T + w.t) * (d.nh * d.nw) + w.win;
  const float* raw = d.rawchan + grp * d.C;
  for (int e = tid; e < w.P; e += 256) {
    const int el = chan_elem(w, e);
    es[e] = el;
    gs[e] = dcx[((int64_t)w.b * d.T + w.t) * d.ce + el];
  }
  float mx = -INFINITY;
  for (int c = tid; c < d.C; c += 256) { const float v = raw[c] * d.scale; pr[c] = v; mx = fmaxf(mx, v); }
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float l = 0.f;
  for (int c = tid; c < d.C; c += 256) { const float pv = expf(pr[c] - mx); pr[c] = pv; l += pv; }
  l = wave_sum(l);
  if (lane == 0) red[4 + wave] = l;
  __syncthreads();
  const float inv = 1.0f / ((red[4] + red[5]) + (red[6] + red[7]));
  const int64_t kvb = (int64_t)w.b * 2 * d.ce * d.ldk;
  float* dS = ws + grp * d.C;
  float* Pw = ws + (int64_t)d.B * d.T * d.nh * d.nw * d.C + grp * d.C;
  float dot = 0.f;
  for (int c = tid; c < d.C; c += 256) {
    float dp = 0.f;
    for (int e = 0; e < w.P; ++e) dp = fmaf(gs[e], ld_elem(d.kvT, kvb + (int64_t)(d.ce + es[e]) * d.ldk + c, d.kv_dtype), dp);
    const float pv = pr[c] * inv;
    Pw[c] = pv;
    dS[c] = dp;                                     // dP for now
    dot += pv * dp;
  }
  __syncthreads();
  dot = wave_sum(dot);
  if (lane == 0) red[wave] = dot;
  __syncthreads();
  dot = (red[0] + red[1]) + (red[2] + red[3]);
  for (int c = tid; c < d.C; c += 256) dS[c] = d.scale * Pw[c] * (dS[c] - dot) + (drawchan ? drawchan[grp * d.C + c] : 0.f);
}
__global__ __launch_bounds__(256) void chanattn_bwd_q_kernel(const mtt_chanattn_desc d, const float* ws, float* dq) {
  const ChanWin w = chan_win(d, blockIdx.y);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t grp = ((int64_t)w.b * d.T + w.t) * (d.nh * d.nw) + w.win;
  const float* dS = ws + grp * d.C;
  const int64_t kvb = (int64_t)w.b * 2 * d.ce * d.ldk;
  for (int k = 0; k < 8; ++k) {
    const int e = blockIdx.x * 32 + wave * 8 + k;
    if (e >= w.P) break;
    const int el = chan_elem(w, e);
    float a = 0.f;
    for (int c = lane; c < d.C; c += 64) a = fmaf(dS[c], ld_elem(d.kvT, kvb + (int64_t)el * d.ldk + c, d.kv_dtype), a);
    a = wave_sum(a);
    if (lane == 0) dq[((int64_t)w.b * d.T + w.t) * d.ce + el] = a;
  }
}
__global__ __launch_bounds__(256) void chanattn_bwd_kv_kernel(const mtt_chanattn_desc d, const float* dcx, const float* ws, float* dkvT, int64_t ldg) {
  const int nwin = d.nh * d.nw;
  const int r = (int)(sqrtf((float)d.ce) + 0.5f), wh = r / d.nh, ww = r / d.nw;
  const float* Pw = ws + (int64_t)d.B * d.T * nwin * d.C;
  const int64_t total = (int64_t)d.B * d.ce * d.C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % d.C);
    const int el = (int)((i / d.C) % d.ce);
    const int b = (int)(i / ((int64_t)d.C * d.ce));
    const int win = ((el / r) / wh) * d.nw + (el % r) / ww;
    float gk = 0.f, gv = 0.f;
    for (int t = 0; t < d.T; ++t) {
      const int64_t grp = ((int64_t)b * d.T + t) * nwin + win;
      gk = fmaf(d.q[((int64_t)b * d.T + t) * d.ce + el], ws[grp * d.C + c], gk);
      gv = fmaf(dcx[((int64_t)b * d.T + t) * d.ce + el], Pw[grp * d.C + c], gv);
    }
    dkvT[((int64_t)b * 2 * d.ce + el) * ldg + c] = gk;
    dkvT[((int64_t)b * 2 * d.ce + d.ce + el) * ldg + c] = gv;
  }
}

// ------------------------------------------------------------------------------------------------
// Backward of conv3s2_kernel (autograd of PatchMerging.spa_attn_ds, taskprompter_swin.py:437,463): input gradient by gathering
// (a thread per input element), weight gradient with one workgroup per (co, ci) pair reducing the 9 taps over all output pixels
// through LDS in a fixed order, bias gradient with one workgroup per output channel.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv3s2_bwd_dx_kernel(const mtt_conv3s2_desc d, const float* dy, float* dx) {
  const int Ho = d.H / 2, Wo = d.W / 2;
  const int64_t total = (int64_t)d.B * d.Ci * d.H * d.W;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int xx = (int)(t % d.W), yy = (int)((t / d.W) % d.H), ci = (int)((t / ((int64_t)d.W * d.H)) % d.Ci);
    const int64_t b = t / ((int64_t)d.W * d.H * d.Ci);
    float a = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int ty = yy + 1 - ky;
      if (ty < 0 || (ty & 1) || (ty >> 1) >= Ho) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int tx = xx + 1 - kx;
        if (tx < 0 || (tx & 1) || (tx >> 1) >= Wo) continue;
        const float* gp = dy + b * d.y_bs + d.y_off + (int64_t)(ty >> 1) * Wo + (tx >> 1);
        for (int co = 0; co < d.Co; ++co) a = fmaf(gp[(int64_t)co * d.y_cs], d.w[((int64_t)co * d.Ci + ci) * 9 + ky * 3 + kx], a);
      }
    }
    dx[b * d.x_bs + (int64_t)ci * d.x_cs + d.x_off + (int64_t)yy * d.W + xx] = a;
  }
}
__global__ __launch_bounds__(256) void conv3s2_bwd_dw_kernel(const mtt_conv3s2_desc d, const float* dy, float* dw, float* db) {
  __shared__ float red[256];
  const int Ho = d.H / 2, Wo = d.W / 2;
  const int co = blockIdx.x / d.Ci, ci = blockIdx.x % d.Ci;
  const int64_t npix = (int64_t)d.B * Ho * Wo;
  float acc[9], sb = 0.f;
#pragma unroll
  for (int k = 0; k < 9; ++k) acc[k] = 0.f;
  for (int64_t i = threadIdx.x; i < npix; i += 256) {
    const int xo = (int)(i % Wo), yo = (int)((i / Wo) % Ho);
    const int64_t b = i / ((int64_t)Wo * Ho);
    const float g = dy[b * d.y_bs + (int64_t)co * d.y_cs + d.y_off + (int64_t)yo * Wo + xo];
    sb += g;
    const float* xp = d.x + b * d.x_bs + (int64_t)ci * d.x_cs + d.x_off;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int yy = 2 * yo - 1 + ky;
      if (yy < 0 || yy >= d.H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int xx = 2 * xo - 1 + kx;
        if (xx < 0 || xx >= d.W) continue;
        acc[ky * 3 + kx] = fmaf(g, xp[(int64_t)yy * d.W + xx], acc[ky * 3 + kx]);
      }
    }
  }
  for (int k = 0; k < 10; ++k) {                  // 9 taps + (ci == 0: the bias gradient of channel co), fixed-order LDS tree
    if (k == 9 && (ci != 0 || !db)) break;
    red[threadIdx.x] = k < 9 ? acc[k] : sb;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) { if (k < 9) dw[((int64_t)co * d.Ci + ci) * 9 + k] = red[0]; else db[co] = red[0]; }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// Bilinear resize of fp32 planes (align_corners = False): the img_ds_ratio input resize (taskprompter_swin.py:666-667).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void resize_nchw_kernel(const float* in, float* out, int planes, int Hin, int Win, int Hout, int Wout) {
  const float sy = (float)Hin / (float)Hout, sx = (float)Win / (float)Wout;
  const int64_t total = (int64_t)planes * Hout * Wout;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int xo = (int)(t % Wout), yo = (int)((t / Wout) % Hout);
    const int64_t pl = t / ((int64_t)Wout * Hout);
    float fy = ((float)yo + 0.5f) * sy - 0.5f; if (fy < 0.f) fy = 0.f;
    float fx = ((float)xo + 0.5f) * sx - 0.5f; if (fx < 0.f) fx = 0.f;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < Hin - 1 ? 1 : 0), x1 = x0 + (x0 < Win - 1 ? 1 : 0);
    const float wy = fy - (float)y0, wx = fx - (float)x0;
    const float* p = in + pl * Hin * Win;
    const float top = p[(int64_t)y0 * Win + x0] * (1.f - wx) + p[(int64_t)y0 * Win + x1] * wx;
    const float bot = p[(int64_t)y1 * Win + x0] * (1.f - wx) + p[(int64_t)y1 * Win + x1] * wx;
    out[t] = top * (1.f - wy) + bot * wy;
  }
}

int grid_for(int64_t n) {
  int64_t g = (n + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 65535 * 4 ? 65535 * 4 : g));
}

}  // namespace

#define S_ ((hipStream_t)stream)

extern "C" int mtt_patchify(const float* img, void* cols, int B, int H, int W, int P, int64_t ldc, int out_dtype, void* stream) {
  if (!img || !cols || B <= 0 || P <= 0 || (H % P) || (W % P) || ldc < 3 * P * P) return MTT_E_BADARG;
  const int64_t n = (int64_t)B * (H / P) * (W / P) * (3 * P + 1);
  if (P == 2) hipLaunchKernelGGL(patchify_p_kernel<2>, dim3(grid_for(n)), dim3(256), 0, S_, img, cols, B, H, W, ldc, out_dtype);
  else if (P == 4) hipLaunchKernelGGL(patchify_p_kernel<4>, dim3(grid_for(n)), dim3(256), 0, S_, img, cols, B, H, W, ldc, out_dtype);
  else if (P == 8) hipLaunchKernelGGL(patchify_p_kernel<8>, dim3(grid_for(n)), dim3(256), 0, S_, img, cols, B, H, W, ldc, out_dtype);
  else return MTT_E_UNSUPPORTED;
  return (int)hipGetLastError();
}

extern "C" int mtt_resize_nchw(const float* in, float* out, int planes, int Hin, int Win, int Hout, int Wout, void* stream) {
  if (!in || !out || planes <= 0 || Hin <= 0 || Win <= 0 || Hout <= 0 || Wout <= 0) return MTT_E_BADARG;
  hipLaunchKernelGGL(resize_nchw_kernel, dim3(grid_for((int64_t)planes * Hout * Wout)), dim3(256), 0, S_, in, out, planes, Hin, Win, Hout, Wout);
  return (int)hipGetLastError();
}

extern "C" int mtt_gather_rows(const mtt_gather_desc* d, void* stream) {
  if (!d || !d->src || !d->dst || !d->idx || d->rows <= 0 || d->B <= 0 || d->C <= 0 || (d->C % 8)) return MTT_E_BADARG;
  if ((d->ld_src % 8) || (d->ld_dst % 8) || (d->src_bs % 8) || (d->dst_bs % 8) || ((uintptr_t)d->src & 15) || ((uintptr_t)d->dst & 15)) return MTT_E_ALIGN;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for((int64_t)d->B * d->rows * (d->C / 8))), dim3(256), 0, S_, *d);
  return (int)hipGetLastError();
}

namespace {

// ------------------------------------------------------------------------------------------------
// Window attention on fp32 storage with matrix-core arithmetic (mtt_winattn_desc.mfma = 1, round 6): the structure of winattn_bf16_kernel
// with every product as three bf16 MFMAs on hi / lo split operands (Qh Kh + Qh Kl + Ql Kh; Ph Vh + Ph Vl + Pl Vh; fp32 accumulate,
// fp32-class products like the x3 GEMMs).  Q / K rows are split in registers from 32-byte fp32 loads, V^T goes to LDS as two planes,
// P is split in registers.  Replaces the one-thread-per-query exact VALU kernel in the x3f / x3 forward (9 % of the Swin-B x3f step).
// ------------------------------------------------------------------------------------------------
MTT_DEV void st_row(bf16_t* p, u32x4 v) { *(u32x2*)p = (u32x2){v[0], v[1]}; *(u32x2*)(p + 4) = (u32x2){v[2], v[3]}; }
MTT_DEV u32x4 ld_row(const bf16_t* p) { const u32x2 a = *(const u32x2*)p, b = *(const u32x2*)(p + 4); return (u32x4){a[0], a[1], b[0], b[1]}; }
template <int NKT>
__global__ __launch_bounds__(256, 2) void winattn_x3_kernel(const mtt_winattn_desc d) {
  constexpr int NP = NKT * 16, PITCH = NP + 8, RP = 36;
  __shared__ __attribute__((aligned(16))) bf16_t vTh[32 * PITCH];
  __shared__ __attribute__((aligned(16))) bf16_t vTl[32 * PITCH];
  __shared__ __attribute__((aligned(16))) bf16_t kRh[NP * RP];         // K rows split ONCE per workgroup (every query tile re-read and re-split them
  __shared__ __attribute__((aligned(16))) bf16_t kRl[NP * RP];         // from L2 before: 10 x the loads and the VALU split work)
  const int nH = d.nH, T = d.T, ws2 = d.ws2, N = T + ws2, C = nH * 32, ld = 3 * C;
  const int win = blockIdx.x / nH, h = blockIdx.x % nH;
  const int wl = win % d.nW, b = win / d.nW;
  const float* base = (const float*)d.qkv + (int64_t)win * N * ld + h * 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;

  for (int key = tid; key < NP; key += 256) {          // (a lane per (row, 8-dim chunk) instead — as in the backward's prologue — measured 3 % slower here)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      u32x4 hi, lo;
      load8<true>(base, C + key * ld + q * 8, MTT_F32, key < N, hi, lo);
      st_row(kRh + key * RP + q * 8, hi); st_row(kRl + key * RP + q * 8, lo);
      load8<true>(base, 2 * C + key * ld + q * 8, MTT_F32, key < N, hi, lo);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        vTh[(q * 8 + 2 * e) * PITCH + key] = (bf16_t)(hi[e] & 0xffffu);
        vTh[(q * 8 + 2 * e + 1) * PITCH + key] = (bf16_t)(hi[e] >> 16);
        vTl[(q * 8 + 2 * e) * PITCH + key] = (bf16_t)(lo[e] & 0xffffu);
        vTl[(q * 8 + 2 * e + 1) * PITCH + key] = (bf16_t)(lo[e] >> 16);
      }
    }
  }
  __syncthreads();

  const float* bias_h = d.bias + (int64_t)h * ws2 * ws2;
  const float* mask_w = d.mask ? d.mask + (int64_t)wl * ws2 * ws2 : nullptr;
  const int nqt = (N + 15) / 16;
  const f32x4 z4 = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int qt = wave; qt < nqt; qt += 4) {
    const int query = qt * 16 + li;
    u32x4 qh, ql;
    load8<true>(base, query * ld + lg * 8, MTT_F32, query < N, qh, ql);
    f32x4 s[NKT];
#pragma unroll
    for (int j = 0; j < NKT; ++j) {
      const int krow = j * 16 + li;
      const u32x4 kh = ld_row(kRh + krow * RP + lg * 8), kl = ld_row(kRl + krow * RP + lg * 8);
      s[j] = mfma16(kh, qh, mfma16(kh, ql, mfma16(kl, qh, z4)));
    }
    if (qt == 0 && li < T && d.rawmap) {
      float* row = d.rawmap + (((int64_t)b * nH + h) * T + li) * d.map_ld + d.map_off;
      const int32_t* px = d.pix + (int64_t)wl * ws2;
#pragma unroll
      for (int j = 0; j < NKT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = j * 16 + lg * 4 + r;
          if (key >= T && key < N) {
            const int pp = px[key - T];
            if (pp >= 0) row[pp] = s[j][r];
          }
        }
    }
    const bool qwin = query >= T && query < N;
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < NKT; ++j) {
      const int kb = j * 16 + lg * 4;
      float add[4] = {0.f, 0.f, 0.f, 0.f};
      if (qwin && kb >= T && kb + 3 < N) {
        const int o = (query - T) * ws2 + (kb - T);
        const F4u bb = *(const F4u*)(bias_h + o);
        add[0] = bb.x; add[1] = bb.y; add[2] = bb.z; add[3] = bb.w;
        if (mask_w) { const F4u mm = *(const F4u*)(mask_w + o); add[0] += mm.x; add[1] += mm.y; add[2] += mm.z; add[3] += mm.w; }
      } else if (qwin) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = kb + r;
          if (key >= T && key < N) {
            const int o = (query - T) * ws2 + (key - T);
            add[r] = bias_h[o] + (mask_w ? mask_w[o] : 0.f);
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = s[j][r] * d.scale + add[r];
        if (kb + r >= N) v = -INFINITY;
        s[j][r] = v;
        mx = fmaxf(mx, v);
      }
    }
    mx = groups_max(mx);
    float l = 0.f;
#pragma unroll
    for (int j = 0; j < NKT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pv = expf(s[j][r] - mx);
        s[j][r] = pv;
        l += pv;
      }
    l = groups_sum(l);
    f32x4 o[2] = {z4, z4};
#pragma unroll
    for (int c = 0; c < NKT / 2; ++c) {
      const f32x4 a = s[2 * c], e = s[2 * c + 1];
      const u32x4 ph = (u32x4){pack2(a[0], a[1]), pack2(a[2], a[3]), pack2(e[0], e[1]), pack2(e[2], e[3])};
      const u32x4 pl = (u32x4){pack2(a[0] - lo_of(ph[0]), a[1] - hi_of(ph[0])), pack2(a[2] - lo_of(ph[1]), a[3] - hi_of(ph[1])),
                               pack2(e[0] - lo_of(ph[2]), e[1] - hi_of(ph[2])), pack2(e[2] - lo_of(ph[3]), e[3] - hi_of(ph[3]))};
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const int off = (dt * 16 + li) * PITCH + 32 * c + 4 * lg;
        const u32x2 h0 = *(const u32x2*)(vTh + off), h1 = *(const u32x2*)(vTh + off + 16);
        const u32x2 l0 = *(const u32x2*)(vTl + off), l1 = *(const u32x2*)(vTl + off + 16);
        const u32x4 vh = (u32x4){h0[0], h0[1], h1[0], h1[1]}, vl = (u32x4){l0[0], l0[1], l1[0], l1[1]};
        o[dt] = mfma16(vh, ph, mfma16(vh, pl, mfma16(vl, ph, o[dt])));
      }
    }
    if (query < N) {
      const float inv = 1.0f / l;
      float* op = (float*)d.out + ((int64_t)win * N + query) * C + h * 32 + lg * 4;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) *(float4*)(op + dt * 16) = make_float4(o[dt][0] * inv, o[dt][1] * inv, o[dt][2] * inv, o[dt][3] * inv);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Window attention backward on the matrix cores (round 6; bf16 products, fp32 accumulate and softmax algebra): bf16 storage, or fp32
// storage with mtt_winattn_desc.mfma = 1 (the bf16 backward of the x3f mode; operands are rounded while loaded).  One workgroup per
// (window, head), two passes in the forward kernel's swapped-product layout so that the softmax statistics stay per-lane scalars and no
// N x N tile goes through LDS:
//   pass 1, a wave per 16-query tile (lane = one query, 4 keys per tile):  S^T = K Q^T, dP^T = V dO^T (one 16x16x32 MFMA per tile each),
//           row max / 1 / sum / D = dO . O -> LDS,  g = scale P (dP - D) + draw,  dS_out,  dQ^T += K^T g^T with the C layout of two g tiles
//           as the B fragment (K^T from LDS in the matching key permutation);
//   pass 2, a wave per 16-key tile (lane = one key, 4 queries per tile):   S = Q K^T, dP = dO V^T recomputed with the operands swapped,
//           P from the stored statistics,  dV^T += dO^T P,  dK^T += Q^T g  (dO^T, Q^T from LDS).
// 7 MFMAs per 16 x 16 tile pair instead of the exact VALU kernel's 5 x 32 FMAs per score (that kernel: 22 % of the Swin-B x3f step).
// ------------------------------------------------------------------------------------------------
// 8 consecutive elements at p + off (32-bit element offset from a per-workgroup base: scalar base + 32-bit lane offset addressing) as packed
// bf16; branch-free: a row beyond the window reads offset 0 and is zeroed afterwards
template <bool F32>
MTT_DEV u32x4 wa_frag(const void* p, int off, bool ok) {
  Raw8<F32> r;
  off = ok ? off : 0;
  if constexpr (F32) {
    r.v0 = *(const float4*)((const float*)p + off);
    r.v1 = *(const float4*)((const float*)p + off + 4);
  } else {
    r.r0 = *(const u32x4*)((const bf16_t*)p + off);
  }
  u32x4 hi, lo;
  cvt8<false, F32>(ok, r, hi, lo);
  return hi;
}
template <int NKT, bool F32>
__global__ __launch_bounds__(256, 2) void winattn_bwd_mfma_kernel(const mtt_winattn_desc d, const void* dout, const float* drawmap, void* dqkv, float* dS_out) {
  // LDS (79.5 KB at NKT = 10: two workgroups per CU): the four operands row-major as bf16 (rows of 32 + 4 elements: 8-byte fragment
  // reads) — every fragment of both passes comes from here (the first version re-read K / V per query tile and Q / dO per key tile from
  // L2: ~1 MB per workgroup) — plus K^T, Q^T, dO^T for the products whose reduction runs over keys / queries
  constexpr int NP = NKT * 16, PITCH = NP + 4, RP = 36;
  __shared__ __attribute__((aligned(16))) bf16_t kT[32 * PITCH];
  __shared__ __attribute__((aligned(16))) bf16_t qT[32 * PITCH];
  __shared__ __attribute__((aligned(16))) bf16_t gT[32 * PITCH];       // dO^T
  __shared__ __attribute__((aligned(16))) bf16_t qR[NP * RP];
  __shared__ __attribute__((aligned(16))) bf16_t kR[NP * RP];
  __shared__ __attribute__((aligned(16))) bf16_t vR[NP * RP];
  __shared__ __attribute__((aligned(16))) bf16_t gR[NP * RP];          // dO
  __shared__ __attribute__((aligned(16))) float mrow[NP];
  __shared__ __attribute__((aligned(16))) float linv[NP];
  __shared__ __attribute__((aligned(16))) float Drow[NP];
  const int nH = d.nH, T = d.T, ws2 = d.ws2, N = T + ws2, C = nH * 32, ld = 3 * C;
  const int win = blockIdx.x / nH, h = blockIdx.x % nH;
  const int wl = win % d.nW, b = win / d.nW;
  const int64_t base = (int64_t)win * N * ld + h * 32, obase = (int64_t)win * N * C + h * 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int dt_ = F32 ? MTT_F32 : MTT_BF16;
  const int es = F32 ? 4 : 2;
  const void* Qb = (const char*)d.qkv + base * es;            // this (window, head)'s Q rows; K at + C, V at + 2 C elements; row pitch ld
  const void* Gb = (const char*)dout + obase * es;            // dO rows (pitch C)
  const void* Ob = (const char*)d.out + obase * es;           // O rows
  void* Db = (char*)dqkv + base * es;                         // dQ | dK | dV rows

  // one (row, 8-dim chunk) per lane and iteration: 4 NP work items over all 256 lanes (a lane per ROW left 96-110 lanes idle and each
  // active one with 16 loads + 112 LDS stores in a row), 4 consecutive lanes cover a row's 128 contiguous bytes
  for (int it = tid; it < 4 * NP; it += 256) {
    const int row = it >> 2;
    const bool ok = row < N;
    {
      const int q = it & 3;
      const u32x4 k4 = wa_frag<F32>(Qb, C + row * ld + q * 8, ok);
      const u32x4 q4 = wa_frag<F32>(Qb, row * ld + q * 8, ok);
      const u32x4 g4 = wa_frag<F32>(Gb, row * C + q * 8, ok);
      const u32x4 v4 = wa_frag<F32>(Qb, 2 * C + row * ld + q * 8, ok);
      st_row(qR + row * RP + q * 8, q4); st_row(kR + row * RP + q * 8, k4); st_row(vR + row * RP + q * 8, v4); st_row(gR + row * RP + q * 8, g4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        kT[(q * 8 + 2 * e) * PITCH + row] = (bf16_t)(k4[e] & 0xffffu); kT[(q * 8 + 2 * e + 1) * PITCH + row] = (bf16_t)(k4[e] >> 16);
        qT[(q * 8 + 2 * e) * PITCH + row] = (bf16_t)(q4[e] & 0xffffu); qT[(q * 8 + 2 * e + 1) * PITCH + row] = (bf16_t)(q4[e] >> 16);
        gT[(q * 8 + 2 * e) * PITCH + row] = (bf16_t)(g4[e] & 0xffffu); gT[(q * 8 + 2 * e + 1) * PITCH + row] = (bf16_t)(g4[e] >> 16);
      }
    }
  }
  __syncthreads();

  const float* bias_h = d.bias + (int64_t)h * ws2 * ws2;
  const float* biasT_h = d.biasT ? d.biasT + (int64_t)h * ws2 * ws2 : nullptr;
  const float* mask_w = d.mask ? d.mask + (int64_t)wl * ws2 * ws2 : nullptr;
  const int32_t* px = d.pix + (int64_t)wl * ws2;
  const float* draw_b = drawmap ? drawmap + ((int64_t)b * nH + h) * T * d.map_ld + d.map_off : nullptr;
  float* dS_wh = dS_out ? dS_out + ((int64_t)win * nH + h) * ws2 * ws2 : nullptr;
  const int ntile = (N + 15) / 16;
  const f32x4 z4 = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- pass 1: a wave per query tile ----
  for (int qt = wave; qt < ntile; qt += 4) {
    const int query = qt * 16 + li;
    const bool qok = query < N;
    const u32x4 qf = ld_row(qR + query * RP + lg * 8);            // (rows >= N are zeros in LDS)
    const u32x4 gf = ld_row(gR + query * RP + lg * 8);
    float Dq = 0.f;
    if (qok) {
      float g8[8], o8[8];
      ld8g(Gb, query * C + lg * 8, dt_, g8);
      ld8g(Ob, query * C + lg * 8, dt_, o8);
#pragma unroll
      for (int e = 0; e < 8; ++e) Dq = fmaf(g8[e], o8[e], Dq);
    }
    Dq = groups_sum(Dq);
    f32x4 s[NKT];
#pragma unroll
    for (int j = 0; j < NKT; ++j) {
      const int krow = j * 16 + li;
      const u32x4 kf = ld_row(kR + krow * RP + lg * 8);
      s[j] = mfma16(kf, qf, z4);
      if (NKT > 6 && (j & 3) == 3) __builtin_amdgcn_sched_barrier(0);     // bound what the scheduler hoists (the 256-register cap: no spills)
    }
    const bool qwin = query >= T && qok;
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < NKT; ++j) {
      const int kb = j * 16 + lg * 4;
      float add[4] = {0.f, 0.f, 0.f, 0.f};
      if (qwin && kb >= T && kb + 3 < N) {
        const int o = (query - T) * ws2 + (kb - T);                  // 32-bit offsets from uniform bases (no per-tile 64-bit pointers kept live)
        const F4u bb = *(const F4u*)(bias_h + o);
        add[0] = bb.x; add[1] = bb.y; add[2] = bb.z; add[3] = bb.w;
        if (mask_w) { const F4u mm = *(const F4u*)(mask_w + o); add[0] += mm.x; add[1] += mm.y; add[2] += mm.z; add[3] += mm.w; }
      } else if (qwin) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = kb + r;
          if (key >= T && key < N) {
            const int o = (query - T) * ws2 + (key - T);
            add[r] = bias_h[o] + (mask_w ? mask_w[o] : 0.f);
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = s[j][r] * d.scale + add[r];
        if (kb + r >= N) v = -INFINITY;
        s[j][r] = v;
        mx = fmaxf(mx, v);
      }
      if (NKT > 6 && (j & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    mx = groups_max(mx);
    float l = 0.f;
#pragma unroll
    for (int j = 0; j < NKT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pv = __expf(s[j][r] - mx);
        s[j][r] = pv;
        l += pv;
      }
    l = groups_sum(l);
    const float inv = 1.0f / l;
    if (lg == 0) { mrow[query] = mx; linv[query] = qok ? inv : 0.f; Drow[query] = Dq; }
    // g = scale P (dP - D) + draw  (kept in s), dS_out
    const bool has_draw = draw_b && query < T;
    const int drow_off = query * (int)d.map_ld;
    const bool has_ds = dS_wh && qwin;
    const int ds_off = (query - T) * ws2 - T;
#pragma unroll
    for (int j = 0; j < NKT; ++j) {
      const int kb = j * 16 + lg * 4;
      const int vrow = j * 16 + li;                      // dP^T tile j = V_j dO^T, consumed at once (no N-long dP row is kept)
      const u32x4 vf = ld_row(vR + vrow * RP + lg * 8);
      const f32x4 dpj = mfma16(vf, gf, z4);
      float ds[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        ds[r] = s[j][r] * inv * (dpj[r] - Dq);
        float g = d.scale * ds[r];
        if (has_draw) {
          const int key = kb + r;
          if (key >= T && key < N) {
            const int pp = px[key - T];
            if (pp >= 0) g += draw_b[drow_off + pp];
          }
        }
        s[j][r] = qok ? g : 0.f;
      }
      if (has_ds) {
        if (kb >= T && kb + 3 < N) *(F4u*)(dS_wh + (ds_off + kb)) = (F4u){ds[0], ds[1], ds[2], ds[3]};
        else {
#pragma unroll
          for (int r = 0; r < 4; ++r) if (kb + r >= T && kb + r < N) dS_wh[ds_off + kb + r] = ds[r];
        }
      }
      if (NKT > 6 && (j & 1) == 1) __builtin_amdgcn_sched_barrier(0);
    }
    f32x4 o[2] = {z4, z4};
#pragma unroll
    for (int c = 0; c < NKT / 2; ++c) {
      const u32x4 gb = (u32x4){pack2(s[2 * c][0], s[2 * c][1]), pack2(s[2 * c][2], s[2 * c][3]),
                               pack2(s[2 * c + 1][0], s[2 * c + 1][1]), pack2(s[2 * c + 1][2], s[2 * c + 1][3])};
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const bf16_t* kr = kT + (dt * 16 + li) * PITCH + 32 * c + 4 * lg;
        const u32x2 lo = *(const u32x2*)kr, hi = *(const u32x2*)(kr + 16);
        o[dt] = mfma16((u32x4){lo[0], lo[1], hi[0], hi[1]}, gb, o[dt]);
      }
    }
    if (qok) {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const int at = query * ld + dt * 16 + lg * 4;
        if (F32) *(float4*)((float*)Db + at) = make_float4(o[dt][0], o[dt][1], o[dt][2], o[dt][3]);
        else *(u32x2*)((bf16_t*)Db + at) = (u32x2){pack2(o[dt][0], o[dt][1]), pack2(o[dt][2], o[dt][3])};
      }
    }
  }
  __syncthreads();

  // ---- pass 2: a wave per key tile ----
  for (int kt = wave; kt < ntile; kt += 4) {
    const int key = kt * 16 + li;
    const bool kok = key < N;
    const u32x4 kf = ld_row(kR + key * RP + lg * 8);
    const u32x4 vf = ld_row(vR + key * RP + lg * 8);
    f32x4 dv[2] = {z4, z4}, dk[2] = {z4, z4};
#pragma unroll 1
    for (int c = 0; c < NKT / 2; ++c) {                 // (not unrolled: the scheduler would hoist every pair's four fragment loads — spills under the 2-workgroups-per-CU register cap)
      f32x4 pt[2], gt[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int t = 2 * c + u;
        const int qrow = t * 16 + li;
        const u32x4 qa = ld_row(qR + qrow * RP + lg * 8);
        const u32x4 ga = ld_row(gR + qrow * RP + lg * 8);
        const f32x4 sv = mfma16(qa, kf, z4);
        const f32x4 dpv = mfma16(ga, vf, z4);
        const int q0 = t * 16 + lg * 4;
        const float4 m4 = *(const float4*)(mrow + q0), l4 = *(const float4*)(linv + q0), D4 = *(const float4*)(Drow + q0);
        const float mm[4] = {m4.x, m4.y, m4.z, m4.w}, ll[4] = {l4.x, l4.y, l4.z, l4.w}, DD[4] = {D4.x, D4.y, D4.z, D4.w};
        // bias + mask of this lane's (4 consecutive queries, one key): with the TRANSPOSED bias table (mtt_winattn_desc.biasT, ABI 13) one 16-byte
        // load each (the shift mask is symmetric), else four strided scalar loads each
        float add[4] = {0.f, 0.f, 0.f, 0.f};
        const bool vec = biasT_h && kok && key >= T && q0 >= T && q0 + 3 < N;
        if (vec) {
          const int o = (key - T) * ws2 + (q0 - T);
          const F4u bb = *(const F4u*)(biasT_h + o);
          add[0] = bb.x; add[1] = bb.y; add[2] = bb.z; add[3] = bb.w;
          if (mask_w) { const F4u mm4 = *(const F4u*)(mask_w + o); add[0] += mm4.x; add[1] += mm4.y; add[2] += mm4.z; add[3] += mm4.w; }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int q = q0 + r;
          float pv = 0.f, g = 0.f;
          if (q < N && kok) {
            float v = sv[r] * d.scale + add[r];
            if (!vec && q >= T && key >= T) {
              const int o = (q - T) * ws2 + (key - T);
              v += bias_h[o];
              if (mask_w) v += mask_w[o];
            }
            pv = __expf(v - mm[r]) * ll[r];
            g = d.scale * pv * (dpv[r] - DD[r]);
            if (draw_b && q < T && key >= T) {
              const int pp = px[key - T];
              if (pp >= 0) g += draw_b[q * (int)d.map_ld + pp];
            }
          }
          pt[u][r] = pv; gt[u][r] = g;
        }
      }
      const u32x4 pb = (u32x4){pack2(pt[0][0], pt[0][1]), pack2(pt[0][2], pt[0][3]), pack2(pt[1][0], pt[1][1]), pack2(pt[1][2], pt[1][3])};
      const u32x4 gb = (u32x4){pack2(gt[0][0], gt[0][1]), pack2(gt[0][2], gt[0][3]), pack2(gt[1][0], gt[1][1]), pack2(gt[1][2], gt[1][3])};
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const int off = (dt * 16 + li) * PITCH + 32 * c + 4 * lg;
        const u32x2 g0 = *(const u32x2*)(gT + off), g1 = *(const u32x2*)(gT + off + 16);
        const u32x2 q0_ = *(const u32x2*)(qT + off), q1_ = *(const u32x2*)(qT + off + 16);
        dv[dt] = mfma16((u32x4){g0[0], g0[1], g1[0], g1[1]}, pb, dv[dt]);
        dk[dt] = mfma16((u32x4){q0_[0], q0_[1], q1_[0], q1_[1]}, gb, dk[dt]);
      }
    }
    if (kok) {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const int ak = C + key * ld + dt * 16 + lg * 4, av = ak + C;
        if (F32) {
          *(float4*)((float*)Db + ak) = make_float4(dk[dt][0], dk[dt][1], dk[dt][2], dk[dt][3]);
          *(float4*)((float*)Db + av) = make_float4(dv[dt][0], dv[dt][1], dv[dt][2], dv[dt][3]);
        } else {
          *(u32x2*)((bf16_t*)Db + ak) = (u32x2){pack2(dk[dt][0], dk[dt][1]), pack2(dk[dt][2], dk[dt][3])};
          *(u32x2*)((bf16_t*)Db + av) = (u32x2){pack2(dv[dt][0], dv[dt][1]), pack2(dv[dt][2], dv[dt][3])};
        }
      }
    }
  }
}

}  // namespace

extern "C" int mtt_winattn_fwd(const mtt_winattn_desc* d, void* stream) {
  if (!d || !d->qkv || !d->out || !d->bias || !d->pix || d->nwin <= 0 || d->nW <= 0 || d->nH <= 0 || d->T < 0 || d->ws2 <= 0) return MTT_E_BADARG;
  if (d->nwin % d->nW) return MTT_E_BADARG;
  const int N = d->T + d->ws2;
  if (N > 160 || d->T > 16) return MTT_E_UNSUPPORTED;
  const dim3 grid((unsigned)(d->nwin * d->nH));
  if (d->dtype == MTT_F32 && d->mfma) {
    if (((uintptr_t)d->qkv & 15) || ((uintptr_t)d->out & 15)) return MTT_E_ALIGN;
    if (N <= 32) hipLaunchKernelGGL(winattn_x3_kernel<2>, grid, dim3(256), 0, S_, *d);
    else if (N <= 64) hipLaunchKernelGGL(winattn_x3_kernel<4>, grid, dim3(256), 0, S_, *d);
    else if (N <= 96) hipLaunchKernelGGL(winattn_x3_kernel<6>, grid, dim3(256), 0, S_, *d);
    else hipLaunchKernelGGL(winattn_x3_kernel<10>, grid, dim3(256), 0, S_, *d);
    return (int)hipGetLastError();
  }
  if (d->dtype == MTT_F32) {
    const int smem = N * 33 * 4 * 2;
    hipLaunchKernelGGL(winattn_f32_kernel, grid, dim3(256), smem, S_, *d);
    return (int)hipGetLastError();
  }
  if (((uintptr_t)d->qkv & 15) || ((uintptr_t)d->out & 7)) return MTT_E_ALIGN;
  if (N <= 32) hipLaunchKernelGGL(winattn_bf16_kernel<2>, grid, dim3(256), 0, S_, *d);
  else if (N <= 64) hipLaunchKernelGGL(winattn_bf16_kernel<4>, grid, dim3(256), 0, S_, *d);
  else if (N <= 96) hipLaunchKernelGGL(winattn_bf16_kernel<6>, grid, dim3(256), 0, S_, *d);
  else hipLaunchKernelGGL(winattn_bf16_kernel<10>, grid, dim3(256), 0, S_, *d);
  return (int)hipGetLastError();
}

extern "C" int mtt_winattn_bwd(const mtt_winattn_desc* d, const void* dout, const float* drawmap, void* dqkv, float* dS_out, void* stream) {
  if (!d || !d->qkv || !d->out || !d->bias || !d->pix || !dout || !dqkv || d->nwin <= 0 || d->nW <= 0 || d->nH <= 0 || d->T < 0 || d->ws2 <= 0) return MTT_E_BADARG;
  if (d->nwin % d->nW) return MTT_E_BADARG;
  const int N = d->T + d->ws2;
  if (N > 160) return MTT_E_UNSUPPORTED;
  const dim3 grid((unsigned)(d->nwin * d->nH));
  if (d->dtype == MTT_BF16 || d->mfma) {               // matrix-core backward: bf16 storage, or fp32 storage in the x3f mode's bf16 backward
    if (((uintptr_t)d->qkv & 15) || ((uintptr_t)d->out & 15) || ((uintptr_t)dout & 15) || ((uintptr_t)dqkv & 15)) return MTT_E_ALIGN;
#define MTT_WB(NKT_) do { if (d->dtype == MTT_F32) hipLaunchKernelGGL((winattn_bwd_mfma_kernel<NKT_, true>), grid, dim3(256), 0, S_, *d, dout, drawmap, dqkv, dS_out); \
                          else hipLaunchKernelGGL((winattn_bwd_mfma_kernel<NKT_, false>), grid, dim3(256), 0, S_, *d, dout, drawmap, dqkv, dS_out); } while (0)
    if (N <= 32) MTT_WB(2); else if (N <= 64) MTT_WB(4); else if (N <= 96) MTT_WB(6); else MTT_WB(10);
#undef MTT_WB
    return (int)hipGetLastError();
  }
  const int smem = (5 * N * 33 + 3 * N) * 4;
  static std::atomic<unsigned long long> done{0};
  if (int e = mtt_ensure_dyn_lds((const void*)winattn_bwd_kernel, smem, done)) return e;
  hipLaunchKernelGGL(winattn_bwd_kernel, dim3((unsigned)(d->nwin * d->nH)), dim3(256), smem, S_, *d, dout, drawmap, dqkv, dS_out);
  return (int)hipGetLastError();
}

extern "C" int mtt_chanattn_fwd(const mtt_chanattn_desc* d, void* stream) {
  if (!d || !d->q || !d->kvT || !d->rawchan || !d->cx || d->B <= 0 || d->T <= 0 || d->C <= 0 || d->ce <= 0 || d->nh <= 0 || d->nw <= 0) return MTT_E_BADARG;
  const int r = (int)(sqrt((double)d->ce) + 0.5);
  if (r * r != d->ce || (r % d->nh) || (r % d->nw) || (r / d->nh) * (r / d->nw) > 1024 || d->C > 16384) return MTT_E_BADARG;
  const int groups = d->B * d->T * d->nh * d->nw, P = (r / d->nh) * (r / d->nw);
  hipLaunchKernelGGL(chanattn_logits_kernel, dim3((unsigned)((d->C + 255) / 256), (unsigned)groups), dim3(256), 0, S_, *d);
  hipLaunchKernelGGL(chanattn_mix_kernel, dim3((unsigned)((P + 31) / 32), (unsigned)groups), dim3(256), d->C * 4, S_, *d);
  return (int)hipGetLastError();
}

extern "C" size_t mtt_chanattn_bwd_ws_floats(const mtt_chanattn_desc* d) {
  return d ? (size_t)2 * d->B * d->T * d->nh * d->nw * d->C : 0;
}
extern "C" int mtt_chanattn_bwd(const mtt_chanattn_desc* d, const float* drawchan, const float* dcx, float* dq, float* dkvT, int64_t ldg, float* ws,
                                void* stream) {
  if (!d || !d->q || !d->kvT || !d->rawchan || !dcx || !dq || !dkvT || !ws || d->B <= 0 || d->T <= 0 || d->C <= 0 || d->ce <= 0 || d->nh <= 0 || d->nw <= 0)
    return MTT_E_BADARG;
  if (d->kvbias) return MTT_E_UNSUPPORTED;        /* the training path folds the biases into kvT */
  const int r = (int)(sqrt((double)d->ce) + 0.5);
  if (r * r != d->ce || (r % d->nh) || (r % d->nw) || (r / d->nh) * (r / d->nw) > 1024 || d->C > 12288 || ldg < d->C) return MTT_E_BADARG;
  const int groups = d->B * d->T * d->nh * d->nw, P = (r / d->nh) * (r / d->nw);
  hipLaunchKernelGGL(chanattn_bwd_a_kernel, dim3((unsigned)groups), dim3(256), d->C * 4, S_, *d, drawchan, dcx, ws);
  hipLaunchKernelGGL(chanattn_bwd_q_kernel, dim3((unsigned)((P + 31) / 32), (unsigned)groups), dim3(256), 0, S_, *d, (const float*)ws, dq);
  hipLaunchKernelGGL(chanattn_bwd_kv_kernel, dim3(grid_for((int64_t)d->B * d->ce * d->C)), dim3(256), 0, S_, *d, dcx, (const float*)ws, dkvT, ldg);
  return (int)hipGetLastError();
}

extern "C" int mtt_conv3s2_nchw_bwd(const mtt_conv3s2_desc* d, const float* dy, float* dx, float* dw, float* db, void* stream) {
  if (!d || !d->x || !d->w || !dy || d->B <= 0 || d->Ci <= 0 || d->Co <= 0 || d->H <= 0 || d->W <= 0 || (d->H % 2) || (d->W % 2)) return MTT_E_BADARG;
  if (dx) hipLaunchKernelGGL(conv3s2_bwd_dx_kernel, dim3(grid_for((int64_t)d->B * d->Ci * d->H * d->W)), dim3(256), 0, S_, *d, dy, dx);
  if (dw) hipLaunchKernelGGL(conv3s2_bwd_dw_kernel, dim3((unsigned)(d->Co * d->Ci)), dim3(256), 0, S_, *d, dy, dw, db);
  return (int)hipGetLastError();
}

extern "C" int mtt_conv3s2_nchw(const mtt_conv3s2_desc* d, void* stream) {
  if (!d || !d->x || !d->w || !d->y || d->B <= 0 || d->Ci <= 0 || d->Co <= 0 || d->H <= 0 || d->W <= 0 || (d->H % 2) || (d->W % 2)) return MTT_E_BADARG;
  hipLaunchKernelGGL(conv3s2_kernel, dim3(grid_for((int64_t)d->B * d->Co * (d->H / 2) * (d->W / 2))), dim3(256), 0, S_, *d);
  return (int)hipGetLastError();
}
