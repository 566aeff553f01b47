// Host cost and device-side spacing of dependent kernel launches on this box: plain stream launches (NULL stream, created
// stream) against one hipGraph of the same launches.  Build: hipcc --offload-arch=gfx950 -O2 launch_cost.hip -o launch_cost.out
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void spin_kernel(float* p, int iters) {
  float v = p[threadIdx.x & 63];
  for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
  if (v == 123.456f) p[0] = v;
}
struct Big { float* p[60]; int n[60]; };
__global__ void big_arg_kernel(Big b, int iters) {
  float v = b.p[0][threadIdx.x & 63];
  for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
  if (v == 123.456f) b.p[0][0] = v;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  float* d;
  hipMalloc(&d, 1 << 20);
  hipMemset(d, 0, 1 << 20);
  hipStream_t s;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int N = 400;
  for (int iters : {0, 20000}) {                       // empty kernels; ~20 us kernels
    for (int which = 0; which < 2; ++which) {
      hipStream_t st = which ? s : nullptr;
      for (int rep = 0; rep < 2; ++rep) {
        hipDeviceSynchronize();
        hipEventRecord(e0, st);
        double t0 = now();
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(768), 65536, st, d, iters);
        double t1 = now();
        hipEventRecord(e1, st);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("iters %5d  %s stream: host %.2f us per launch, device %.2f us per kernel\n", iters, which ? "created" : "NULL   ", (t1 - t0) / N * 1e6, ms / N * 1e3);
      }
    }
    // the same launches captured into one graph
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(768), 65536, s, d, iters);
    hipStreamEndCapture(s, &g);
    hipError_t r = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    if (r != hipSuccess) { printf("instantiate failed %d\n", (int)r); return 1; }
    for (int rep = 0; rep < 3; ++rep) {
      hipDeviceSynchronize();
      hipEventRecord(e0, s);
      double t0 = now();
      hipGraphLaunch(ge, s);
      double t1 = now();
      hipEventRecord(e1, s);
      hipDeviceSynchronize();
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep) printf("iters %5d  graph of %d:   host %.2f us per node,   device %.2f us per kernel\n", iters, N, (t1 - t0) / N * 1e6, ms / N * 1e3);
    }
    // ... and with one kernel node's parameters replaced before each launch
    {
      std::vector<hipGraphNode_t> nodes(N);
      size_t nn = N;
      hipGraphGetNodes(g, nodes.data(), &nn);
      hipKernelNodeParams kp{};
      hipGraphKernelNodeGetParams(nodes[0], &kp);
      double t0 = now();
      for (int i = 0; i < 100; ++i) hipGraphExecKernelNodeSetParams(ge, nodes[i], &kp);
      double t1 = now();
      printf("hipGraphExecKernelNodeSetParams: %.2f us per node\n", (t1 - t0) / 100 * 1e6);
    }
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
  }
  {
    Big b{};
    for (auto& p : b.p) p = d;
    hipDeviceSynchronize();
    double t0 = now();
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(big_arg_kernel, dim3(256), dim3(256), 0, s, b, 0);
    double t1 = now();
    hipDeviceSynchronize();
    printf("720-byte kernel argument: host %.2f us per launch\n", (t1 - t0) / N * 1e6);
    // events: record + wait pairs
    hipEvent_t ev; hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    hipStream_t s2; hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    t0 = now();
    for (int i = 0; i < N; ++i) { hipEventRecord(ev, s); hipStreamWaitEvent(s2, ev, 0); }
    t1 = now();
    hipDeviceSynchronize();
    printf("hipEventRecord + hipStreamWaitEvent: host %.2f us per pair\n", (t1 - t0) / N * 1e6);
    t0 = now();
    for (int i = 0; i < N; ++i) hipMemsetAsync(d, 0, 16384, s);
    t1 = now();
    hipDeviceSynchronize();
    printf("hipMemsetAsync 16 KiB: host %.2f us\n", (t1 - t0) / N * 1e6);
  }
  return 0;
}
