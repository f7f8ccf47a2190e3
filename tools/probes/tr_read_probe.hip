// Hardware probe (run on the GPU box): lane/element mapping of ds_read_b64_tr_b16 on gfx950.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/tr_read_probe.hip -o /tmp/tr_probe && /tmp/tr_probe
// LDS holds lds[e] = e (16-bit).  Experiment 1: lane l reads at byte address 8*l (its own 4 consecutive elements 4l..4l+3).
// Experiment 2: a [K=16][M=64] row-major bf16 tile (row stride 64 elements); lane l reads at element address
//   (4*(l>>4) + 0)*64 ... i.e. row k = 4*(l>>4) + (l&3)?  -> we print what comes back for addr = ((l & 15) >> 2) * 64 * 1 + ... (see code).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
__global__ void k(unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const unsigned lane = threadIdx.x;
  u32x2 r;
  // experiment 1: contiguous 8-byte pieces
  unsigned addr = (unsigned)(uintptr_t)lds + lane * 8;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
  out[lane * 4 + 0] = r[0] & 0xffff; out[lane * 4 + 1] = r[0] >> 16; out[lane * 4 + 2] = r[1] & 0xffff; out[lane * 4 + 3] = r[1] >> 16;
  // experiment 2: tile [k][m] with row stride 64 elements: lane l -> k-row (l & 15), m-block 4*(l >> 4): address = ((l&15)*64 + 4*(l>>4)) * 2
  addr = (unsigned)(uintptr_t)lds + ((lane & 15) * 64 + 4 * (lane >> 4)) * 2;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
  out[256 + lane * 4 + 0] = r[0] & 0xffff; out[256 + lane * 4 + 1] = r[0] >> 16; out[256 + lane * 4 + 2] = r[1] & 0xffff; out[256 + lane * 4 + 3] = r[1] >> 16;
}
int main() {
  unsigned short* d; unsigned short h[512];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int e = 0; e < 2; ++e) {
    printf("experiment %d: lane -> 4 returned elements (values = LDS element indices)\n", e + 1);
    for (int l = 0; l < 64; ++l) printf("  lane %2d: %4u %4u %4u %4u%s", l, h[e * 256 + l * 4], h[e * 256 + l * 4 + 1], h[e * 256 + l * 4 + 2], h[e * 256 + l * 4 + 3], (l & 3) == 3 ? "\n" : "");
  }
  return 0;
}
