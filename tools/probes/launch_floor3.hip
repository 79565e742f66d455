// Does kernarg preload (-mllvm -amdgpu-kernarg-preload-count=16: the first 16 kernarg dwords arrive in SGPRs with the wave instead of
// by s_load) shorten a dependent chain of short kernels?  Each kernel reads what its predecessor wrote (16 B per thread, 256 WG x 384
// threads) through pointers taken from its kernel arguments -- as scalars (preloadable) or as one by-value struct (byref: never preloaded).
// build twice: hipcc --offload-arch=gfx950 -O3 launch_floor3.hip -o _build/launch_floor3 [-mllvm -amdgpu-kernarg-preload-count=16 -> _kp]
#include <hip/hip_runtime.h>
#include <cstdio>
struct Args { const float4* src; float4* dst; int n; int pad[9]; };
__global__ __launch_bounds__(384) void k_scalar(const float4* src, float4* dst, int n) {
  const int i = blockIdx.x * 384 + threadIdx.x;
  if (i < n) { float4 v = src[i]; v.x += 1.f; dst[i] = v; }
}
__global__ __launch_bounds__(384) void k_struct(Args a) {
  const int i = blockIdx.x * 384 + threadIdx.x;
  if (i < a.n) { float4 v = a.src[i]; v.x += 1.f; a.dst[i] = v; }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <typename F> float time_graph(hipStream_t st, int n, F launch) {
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
  for (int i = 0; i < n; ++i) launch(i);
  hipStreamEndCapture(st, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  for (int r = 0; r < 20; ++r) hipGraphLaunch(ge, st);
  hipStreamSynchronize(st);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a, st);
  for (int r = 0; r < 50; ++r) hipGraphLaunch(ge, st);
  hipEventRecord(b, st); hipStreamSynchronize(st);
  float ms; hipEventElapsedTime(&ms, a, b);
  hipGraphExecDestroy(ge); hipGraphDestroy(g);
  return ms * 1e3f / (50 * n);
}
int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  const int n = 256 * 384;
  float4 *x, *y; CK(hipMalloc(&x, n * 16)); CK(hipMalloc(&y, n * 16)); CK(hipMemset(x, 0, n * 16)); CK(hipMemset(y, 0, n * 16));
  const int N = 400;
  for (int rep = 0; rep < 3; ++rep) {
    float a = time_graph(st, N, [&](int i) { hipLaunchKernelGGL(k_scalar, dim3(256), dim3(384), 0, st, (i & 1) ? y : x, (i & 1) ? x : y, n); });
    float b = time_graph(st, N, [&](int i) { Args s{}; s.src = (i & 1) ? y : x; s.dst = (i & 1) ? x : y; s.n = n; hipLaunchKernelGGL(k_struct, dim3(256), dim3(384), 0, st, s); });
    printf("dependent 16 B-per-thread copy kernels, 256 WG x 384: scalar args %.3f us | struct arg %.3f us per launch\n", a, b);
  }
  return 0;
}
