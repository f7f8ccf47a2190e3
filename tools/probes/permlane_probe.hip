// Hardware probe (run on the GPU box): lane mapping of v_permlane16_swap_b32 / v_permlane32_swap_b32 on gfx950.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/permlane_probe.hip -o /tmp/permlane_probe && /tmp/permlane_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  const unsigned lane = threadIdx.x;
  unsigned a = lane, b = 100 + lane;
  auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[lane] = r[0]; out[64 + lane] = r[1];
  a = lane; b = 100 + lane;
  auto q = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[128 + lane] = q[0]; out[192 + lane] = q[1];
}
int main() {
  unsigned* d; unsigned h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[4] = {"permlane16_swap r[0] (vdst=lane)", "permlane16_swap r[1] (src=100+lane)", "permlane32_swap r[0]", "permlane32_swap r[1]"};
  for (int t = 0; t < 4; ++t) {
    printf("%s:\n", names[t]);
    for (int l = 0; l < 64; ++l) printf("%4u%s", h[t * 64 + l], (l & 15) == 15 ? "\n" : "");
  }
  return 0;
}
