// Per-task training losses straight from the full-resolution logits (TaskPrompter/losses/loss_functions.py:15-177):
//   kind 0  CrossEntropy with ignore index                     (semseg, human_parts)
//   kind 1  2-class CrossEntropy re-weighted by label frequency (sal, `balanced=True`)
//   kind 2  HED-style balanced binary cross entropy             (edge)
//   kind 3  L1 with ignore value                                 (depth)
//   kind 4  L1 on L2-normalised predictions                      (normals)
// pred fp32 NCHW [B, C, HW], label fp32 [B, Cl, HW] (Cl = 1, or = C for kinds 3/4).  Three entry points:
//   mtt_loss_label_stats: stats[0] += #valid pixels, stats[1] += sum of the valid labels (class-frequency weights of kind 1)
//   mtt_loss_fwd        : loss[0] += task loss (normalised with stats, read on the device: no host synchronisation)
//   mtt_loss_bwd        : dpred = gout[0] * d(task loss)/d(pred)
// One thread per pixel: the C logits of a pixel are strided by HW, so a wave reads C coalesced 256-byte rows.
#include "mtt_device.h"

namespace {

constexpr int MAXC = 1 << 20;   // classes are looped at run time

MTT_DEV float block_sum256(float v) {
  __shared__ float red[4];
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  const float r = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  return r;
}

// grid (pixel blocks, B): no per-pixel division; the class loop runs over the C coalesced rows of a pixel block (the second
// read of the backward hits L2: a block touches 256 px x C x 4 B).
__global__ __launch_bounds__(256) void label_stats_kernel(const mtt_loss_desc d, float* part) {
  float cnt = 0.f, pos = 0.f;
  const int64_t b = blockIdx.y;
  const float* lab = d.label + b * d.Cl * d.HW;
  for (int64_t hw = (int64_t)blockIdx.x * 256 + threadIdx.x; hw < d.HW; hw += (int64_t)gridDim.x * 256) {
    bool ok = true;
    for (int c = 0; c < d.Cl; ++c) ok = ok && lab[(int64_t)c * d.HW + hw] != d.ignore;
    if (ok) { cnt += 1.f; pos += lab[hw]; }
  }
  cnt = block_sum256(cnt); pos = block_sum256(pos);
  // per-block partials (block order = blockIdx.y * gridDim.x + blockIdx.x), summed in a fixed order by mtt_reduce_many_kernel
  if (threadIdx.x == 0) { const int64_t bi = (int64_t)blockIdx.y * gridDim.x + blockIdx.x; part[2 * bi] = cnt; part[2 * bi + 1] = pos; }
}

// softplus(-x) = -log(sigmoid(x)), stable
MTT_DEV float softplus_neg(float x) { return fmaxf(-x, 0.f) + log1pf(__expf(-fabsf(x))); }

template <bool BWD>
__global__ __launch_bounds__(256) void loss_kernel(const mtt_loss_desc d, const float* gout) {
  const float nvalid = fmaxf(d.stats[0], 1.f);
  const float inv = 1.0f / nvalid;
  float w0 = 1.f, w1 = 1.f;
  if (d.kind == 1) {                                   // class weights (1 - w_pos, w_pos), w_pos = fraction of zeros among the valid
    const float wpos = (nvalid - d.stats[1]) * inv;
    w0 = 1.f - wpos; w1 = wpos;
  }
  const float factor = 1.0f / (1.0f - d.pos_weight), pw = d.pos_weight * factor;    // kind 2
  const float go = BWD ? gout[0] : 0.f;
  float acc = 0.f;
  const int64_t b = blockIdx.y;
  const float* xb = d.pred + b * d.C * d.HW;
  const float* lab = d.label + b * d.Cl * d.HW;
  float* gb = BWD ? d.dpred + b * d.C * d.HW : nullptr;
  for (int64_t hw = (int64_t)blockIdx.x * 256 + threadIdx.x; hw < d.HW; hw += (int64_t)gridDim.x * 256) {
    const float* x = xb + hw;
    if (d.kind <= 1) {
      const float yl = lab[hw];
      const bool ok = yl != d.ignore;
      const int y = ok ? (int)yl : -1;
      float mx = -INFINITY, se = 0.f, xy = 0.f;      // online log-sum-exp over the classes
#pragma unroll 4
      for (int c = 0; c < d.C; ++c) {
        const float v = x[(int64_t)c * d.HW];
        const float m2 = fmaxf(mx, v);
        se = se * __expf(mx - m2) + __expf(v - m2);
        mx = m2;
        xy = c == y ? v : xy;
      }
      const float wy = d.kind == 1 ? (y == 1 ? w1 : w0) : 1.f;
      if (!BWD) {
        if (ok) acc += wy * (mx + __logf(se) - xy);
      } else {
        const float s = ok ? go * wy * inv : 0.f, rse = 1.0f / se;
        float* g = gb + hw;
#pragma unroll 4
        for (int c = 0; c < d.C; ++c)
          g[(int64_t)c * d.HW] = s * (__expf(x[(int64_t)c * d.HW] - mx) * rse - (c == y ? 1.f : 0.f));
      }
    } else if (d.kind == 2) {
      const float xv = x[0], y = lab[hw];
      const bool ok = y != d.ignore;
      if (!BWD) {
        if (ok) acc += pw * y * softplus_neg(xv) + (1.f - y) * softplus_neg(-xv);
      } else {
        const float sg = 1.0f / (1.0f + __expf(-xv));
        gb[hw] = ok ? go * inv / factor * ((1.f - y) * sg - pw * y * (1.f - sg)) : 0.f;
      }
    } else {
      float v[3], yv[3];
      float nrm = 0.f;
      bool ok = true;
#pragma unroll
      for (int c = 0; c < 3; ++c)
        if (c < d.C) {
          v[c] = x[(int64_t)c * d.HW]; yv[c] = lab[(int64_t)c * d.HW + hw];
          nrm += v[c] * v[c];
          ok = ok && yv[c] != d.ignore;
        }
      nrm = fmaxf(sqrtf(nrm), 1e-12f);
      const float rn = d.kind == 4 ? 1.0f / nrm : 1.f;
      float sgn[3], dot = 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c)
        if (c < d.C) {
          const float o = v[c] * rn, df = o - yv[c];
          if (!BWD && ok) acc += fabsf(df);
          sgn[c] = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);
          dot += o * sgn[c];
        }
      if (BWD) {
        const float s = ok ? go * inv : 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c)
          if (c < d.C) gb[(int64_t)c * d.HW + hw] = d.kind == 4 ? s * (sgn[c] - v[c] * rn * dot) * rn : s * sgn[c];
      }
    }
  }
  if (!BWD) {
    acc = block_sum256(acc);
    if (threadIdx.x == 0) d.ws[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = acc * inv * (d.kind == 2 ? 1.0f / factor : 1.f);
  }
}

// ---- detection-branch losses (det_losses.py): element-wise on [N, C], weights, deterministic sum -----------------------------------
// log(sigmoid(x)) = -softplus(-x), log(1 - sigmoid(x)) = -softplus(x): the stable forms of the reference's BCE-with-logits host path
// (py_sigmoid_focal_loss, det_losses.py:183-224); mmcv's device kernel clamps log at FLT_MIN instead, the same values to fp32 rounding
// wherever |x| < 87.
MTT_DEV void focal_terms(float x, bool pos, float gamma, float alpha, float& loss, float& grad) {
  const float p = 1.0f / (1.0f + __expf(-x));
  const float lp = -softplus_neg(x), ln = -softplus_neg(-x);         // log p, log (1 - p)
  if (pos) {
    const float q = 1.0f - p, m = gamma == 2.0f ? q * q : __powf(fmaxf(q, 1e-38f), gamma);
    loss = -alpha * m * lp;
    // d/dx [-(1-p)^g log p] = (1-p)^g (g p log p - (1 - p))
    grad = alpha * m * (gamma * p * lp - q);
  } else {
    const float m = gamma == 2.0f ? p * p : __powf(fmaxf(p, 1e-38f), gamma);
    loss = -(1.0f - alpha) * m * ln;
    // d/dx [-p^g log(1-p)] = p^g (p - g (1 - p) log(1 - p))
    grad = (1.0f - alpha) * m * (p - gamma * (1.0f - p) * ln);
  }
}

template <bool BWD>
__global__ __launch_bounds__(256) void detloss_kernel(const mtt_detloss_desc d, const float* gscale, const float* gelem, float scale, float* dpred) {
  const int64_t total = d.N * d.C;
  const float gs = BWD ? (gelem ? scale : gscale[0] * scale) : 0.f;
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t n = i / d.C;
    const int c = (int)(i - n * d.C);
    const float x = d.pred[i];
    const float w = d.wmode == 0 ? 1.f : (d.wmode == 1 ? d.weight[n] : d.weight[i]);
    float loss, grad;
    if (d.kind == 0) {
      focal_terms(x, ((const int64_t*)d.target)[n] == (int64_t)c, d.gamma, d.alpha, loss, grad);
    } else {
      const float df = x - ((const float*)d.target)[i], a = fabsf(df);
      loss = a < d.beta ? 0.5f * a * a / d.beta : a - 0.5f * d.beta;
      grad = a < d.beta ? df / d.beta : (df > 0.f ? 1.f : -1.f);
    }
    if (!BWD) {
      const float v = w * loss;
      if (d.out) d.out[i] = v;
      acc += v;
    } else {
      dpred[i] = w * grad * (gelem ? gelem[i] * gs : gs);
    }
  }
  if (!BWD && d.sum) {
    acc = block_sum256(acc);
    if (threadIdx.x == 0) d.ws[blockIdx.x] = acc;
  }
}

unsigned detloss_blocks(const mtt_detloss_desc* d) {
  const int64_t g = (d->N * d->C + 255) / 256;
  return (unsigned)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

int detloss_check(const mtt_detloss_desc* d) {
  if (!d || !d->pred || !d->target || d->N <= 0 || d->C <= 0 || d->kind < 0 || d->kind > 1 || d->wmode < 0 || d->wmode > 2) return MTT_E_BADARG;
  if (d->wmode != 0 && !d->weight) return MTT_E_BADARG;
  if (d->kind == 1 && !(d->beta > 0.f)) return MTT_E_BADARG;
  return 0;
}

dim3 loss_grid(const mtt_loss_desc* d) {
  int64_t g = (d->HW + 255) / 256;
  const int64_t cap = (8192 + d->B - 1) / d->B;
  if (g > cap) g = cap;
  return dim3((unsigned)(g < 1 ? 1 : g), (unsigned)d->B);
}

int loss_check(const mtt_loss_desc* d) {
  if (!d || !d->pred || !d->label || d->B <= 0 || d->HW <= 0 || d->C <= 0 || d->kind < 0 || d->kind > 4) return MTT_E_BADARG;
  if ((d->kind <= 1 && (d->C > MAXC || d->Cl != 1)) || (d->kind == 1 && d->C != 2) || (d->kind == 2 && (d->C != 1 || d->Cl != 1)) ||
      (d->kind >= 3 && (d->C > 3 || d->Cl != d->C)))
    return MTT_E_UNSUPPORTED;
  return 0;
}

}  // namespace

extern "C" size_t mtt_loss_ws_floats(const mtt_loss_desc* d) {
  if (!d || d->B <= 0 || d->HW <= 0) return 0;
  const dim3 g = loss_grid(d);
  return (size_t)2 * g.x * g.y;
}
extern "C" int mtt_loss_label_stats(const mtt_loss_desc* d, float* stats, void* stream) {
  if (!d || !d->label || !stats || !d->ws || d->B <= 0 || d->HW <= 0 || d->Cl <= 0) return MTT_E_BADARG;
  const dim3 g = loss_grid(d);
  hipLaunchKernelGGL(label_stats_kernel, g, dim3(256), 0, (hipStream_t)stream, *d, d->ws);
  hipLaunchKernelGGL(mtt_reduce_many_kernel, dim3(2), dim3(256), 0, (hipStream_t)stream, (const float*)d->ws, (int)(g.x * g.y), 2, stats, 1.0f, 1);
  return (int)hipGetLastError();
}
extern "C" int mtt_loss_fwd(const mtt_loss_desc* d, void* stream) {
  if (int e = loss_check(d)) return e;
  if (!d->stats || !d->loss || !d->ws) return MTT_E_BADARG;
  const dim3 g = loss_grid(d);
  hipLaunchKernelGGL(loss_kernel<false>, g, dim3(256), 0, (hipStream_t)stream, *d, (const float*)nullptr);
  hipLaunchKernelGGL(mtt_reduce_many_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)d->ws, (int)(g.x * g.y), 1, d->loss, 1.0f, 1);
  return (int)hipGetLastError();
}
extern "C" int mtt_loss_bwd(const mtt_loss_desc* d, const float* gout, void* stream) {
  if (int e = loss_check(d)) return e;
  if (!d->stats || !d->dpred || !gout) return MTT_E_BADARG;
  hipLaunchKernelGGL(loss_kernel<true>, loss_grid(d), dim3(256), 0, (hipStream_t)stream, *d, gout);
  return (int)hipGetLastError();
}

extern "C" size_t mtt_detloss_ws_floats(const mtt_detloss_desc* d) {
  if (!d || d->N <= 0 || d->C <= 0) return 0;
  return (size_t)detloss_blocks(d);
}
extern "C" int mtt_detloss_fwd(const mtt_detloss_desc* d, void* stream) {
  if (int e = detloss_check(d)) return e;
  if ((!d->out && !d->sum) || (d->sum && !d->ws)) return MTT_E_BADARG;
  const unsigned g = detloss_blocks(d);
  hipLaunchKernelGGL(detloss_kernel<false>, dim3(g), dim3(256), 0, (hipStream_t)stream, *d, (const float*)nullptr, (const float*)nullptr, 1.0f, (float*)nullptr);
  if (d->sum) hipLaunchKernelGGL(mtt_reduce_many_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)d->ws, (int)g, 1, d->sum, 1.0f, 0);
  return (int)hipGetLastError();
}
extern "C" int mtt_detloss_bwd(const mtt_detloss_desc* d, const float* gscale, const float* gelem, float scale, float* dpred, void* stream) {
  if (int e = detloss_check(d)) return e;
  if (!dpred || (!gscale && !gelem)) return MTT_E_BADARG;
  hipLaunchKernelGGL(detloss_kernel<true>, dim3(detloss_blocks(d)), dim3(256), 0, (hipStream_t)stream, *d, gscale, gelem, scale, dpred);
  return (int)hipGetLastError();
}
