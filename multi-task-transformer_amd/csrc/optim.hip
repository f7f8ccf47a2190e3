// Fused multi-tensor gradient-norm + clip + Adam step (TaskPrompter/utils/train_utils.py:47-51: clip_grad_norm_(max_norm=10)
// followed by torch.optim.Adam.step()).  Two launches for ALL parameters: (1) sum of squares of every gradient,
// (2) Adam update with the clip coefficient applied to the gradient on the fly — 28 B / parameter of HBM traffic
// (grad, param, exp_avg, exp_avg_sq read; param, exp_avg, exp_avg_sq written) instead of clip's extra read + write of
// all gradients.  Tensors are described by device-resident pointer tables; work is cut into fixed-size chunks.
#include "mtt_device.h"

namespace {

constexpr int CHUNK = 65536;   // elements per workgroup (host-side chunk table uses the same constant: mtt_adam_chunk())

MTT_DEV float block_sum(float v) {
  __shared__ float red[4];
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void grad_sqnorm_kernel(const mtt_adam_desc d, float* part) {
  const int t = d.chunk_tensor[blockIdx.x];
  const int64_t off = d.chunk_off[blockIdx.x];
  const float* g = (const float*)d.grads[t] + off;
  const int64_t rem = d.numel[t] - off;
  const int n = rem < CHUNK ? (int)rem : CHUNK;
  float s = 0.f;
  if (((uintptr_t)g & 15) == 0) {
    const int n4 = n >> 2;
    for (int i = threadIdx.x; i < n4; i += 256) {
      const float4 v = ((const float4*)g)[i];
      s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    for (int i = (n4 << 2) + threadIdx.x; i < n; i += 256) s += g[i] * g[i];
  } else {
    for (int i = threadIdx.x; i < n; i += 256) s += g[i] * g[i];
  }
  s = block_sum(s);
  if (threadIdx.x == 0) part[blockIdx.x] = s;          // per-chunk partial; summed in chunk order by mtt_reduce_many_kernel (no atomics)
}

struct AdamHyper { float step_size, inv_sqrt_bc2; };
MTT_DEV void adam1(float g, float& p, float& m, float& v, const mtt_adam_desc& d, float coef, AdamHyper hy) {
  g *= coef;
  if (d.weight_decay != 0.f) g = fmaf(d.weight_decay, p, g);
  m = fmaf(d.beta1, m, (1.0f - d.beta1) * g);
  v = fmaf(d.beta2, v, (1.0f - d.beta2) * g * g);
  const float denom = sqrtf(v) * hy.inv_sqrt_bc2 + d.eps;
  p -= hy.step_size * (m / denom);
}

__global__ __launch_bounds__(256) void adam_step_kernel(const mtt_adam_desc d, const float* total_sq) {
  const int t = d.chunk_tensor[blockIdx.x];
  const int64_t off = d.chunk_off[blockIdx.x];
  const float* g = (const float*)d.grads[t] + off;
  float* p = d.params[t] + off;
  float* m = d.exp_avg[t] + off;
  float* v = d.exp_avg_sq[t] + off;
  const int64_t rem = d.numel[t] - off;
  const int n = rem < CHUNK ? (int)rem : CHUNK;
  float coef = 1.0f;
  if (d.max_norm > 0.f && total_sq) {                     // torch.nn.utils.clip_grad_norm_: min(1, max_norm / (norm + 1e-6))
    coef = d.max_norm / (sqrtf(*total_sq) + 1e-6f);
    coef = coef < 1.0f ? coef : 1.0f;
  }
  const AdamHyper hy = d.hyper ? AdamHyper{d.hyper[0], d.hyper[1]} : AdamHyper{d.step_size, d.inv_sqrt_bc2};
  const bool vec = (((uintptr_t)g | (uintptr_t)p | (uintptr_t)m | (uintptr_t)v) & 15) == 0;
  int done = 0;
  if (vec) {
    const int n4 = n >> 2;
    for (int i = threadIdx.x; i < n4; i += 256) {
      const float4 gv = ((const float4*)g)[i];
      float4 pv = ((float4*)p)[i], mv = ((float4*)m)[i], vv = ((float4*)v)[i];
      adam1(gv.x, pv.x, mv.x, vv.x, d, coef, hy); adam1(gv.y, pv.y, mv.y, vv.y, d, coef, hy);
      adam1(gv.z, pv.z, mv.z, vv.z, d, coef, hy); adam1(gv.w, pv.w, mv.w, vv.w, d, coef, hy);
      ((float4*)p)[i] = pv; ((float4*)m)[i] = mv; ((float4*)v)[i] = vv;
    }
    done = n4 << 2;
  }
  for (int i = done + threadIdx.x; i < n; i += 256) adam1(g[i], p[i], m[i], v[i], d, coef, hy);
}

}  // namespace

extern "C" int mtt_adam_chunk(void) { return CHUNK; }

extern "C" int mtt_grad_sqnorm(const mtt_adam_desc* d, float* out_sq, void* stream) {
  if (!d || !d->grads || !d->numel || !d->chunk_tensor || !d->chunk_off || !out_sq || !d->ws || d->n_chunks <= 0) return MTT_E_BADARG;
  hipLaunchKernelGGL(grad_sqnorm_kernel, dim3(d->n_chunks), dim3(256), 0, (hipStream_t)stream, *d, d->ws);
  hipLaunchKernelGGL(mtt_reduce_many_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)d->ws, d->n_chunks, 1, out_sq, 1.0f, 1);
  return (int)hipGetLastError();
}

extern "C" int mtt_adam_step(const mtt_adam_desc* d, const float* total_sq, void* stream) {
  if (!d || !d->grads || !d->params || !d->exp_avg || !d->exp_avg_sq || !d->numel || !d->chunk_tensor || !d->chunk_off || d->n_chunks <= 0)
    return MTT_E_BADARG;
  hipLaunchKernelGGL(adam_step_kernel, dim3(d->n_chunks), dim3(256), 0, (hipStream_t)stream, *d, total_sq);
  return (int)hipGetLastError();
}
