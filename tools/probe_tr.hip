// Probe: empirical lane/element mapping of ds_read_b64_tr_b16 on gfx950 (run on the GPU box).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  unsigned addr = (unsigned)(size_t)(&lds[0]) + threadIdx.x * 8;   // each lane: its own 8-byte chunk
  uint2 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
  out[threadIdx.x * 4 + 0] = r.x & 0xffff; out[threadIdx.x * 4 + 1] = r.x >> 16;
  out[threadIdx.x * 4 + 2] = r.y & 0xffff; out[threadIdx.x * 4 + 3] = r.y >> 16;
}
int main() {
  unsigned short* d; unsigned short h[256];
  hipMalloc(&d, sizeof h);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
  return 0;
}
