// Micro-benchmark: how fast can W waves per CU (one workgroup per CU, as the persistent GEMMs run) write a GEMM-shaped
// fp32 output with 16-byte stores?  Pattern = the x2 GEMM epilogue's: per store instruction a wave writes 4 rows x 256
// contiguous bytes (row stride N*4).  usage: store_bw [waves ...]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out, int M, int N, int tiles_n, int total) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  const int fi = lane & 15, fg = lane >> 4;
  for (int t = blockIdx.x; t < total; t += gridDim.x) {
    const int m0 = (t / tiles_n) * 256, n0 = (t % tiles_n) * 128;
    // 256 x 128 tile = 64 row-groups of 4 rows x 2 column halves; spread over the waves
    for (int u = wave; u < 128; u += nw) {
      const int rg = u >> 1, ch = u & 1;
      const int m = m0 + rg * 4 + fg, n = n0 + ch * 64 + 4 * fi;
      if (m < M) *reinterpret_cast<f32x4*>(out + (size_t)m * N + n) = (f32x4){(float)t, (float)u, 1.f, 2.f};
    }
  }
}
int main(int argc, char** argv) {
  const int M = 61965, N = 1536, tn = N / 128, tm = (M + 255) / 256, total = tm * tn;
  float* out; hipMalloc(&out, (size_t)M * N * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 1; i < argc; ++i) {
    const int w = atoi(argv[i]);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k, dim3(256), dim3(w * 64), 0, 0, out, M, N, tn, total);
    hipEventRecord(a);
    for (int rep = 0; rep < 10; ++rep) hipLaunchKernelGGL(k, dim3(256), dim3(w * 64), 0, 0, out, M, N, tn, total);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%2d waves/CU: %.1f us per 381 MB pass = %.2f TB/s\n", w, ms * 100, (double)M * N * 4 / (ms * 1e-4) / 1e12);
  }
  return 0;
}
