// Per-launch floor of dependent kernels in a captured graph, by workgroup shape (tools/probes; build: hipcc --offload-arch=gfx950 -O3
// launch_floor.hip -o _build/launch_floor).  Each kernel reads one word written by its predecessor (a real dependency) and exits.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int LDS>
__global__ void k_empty(int* p) {
  __shared__ char pad[LDS > 0 ? LDS : 1];
  if (LDS > 0 && threadIdx.x == 9999) pad[threadIdx.x] = 1;
  if (blockIdx.x == 0 && threadIdx.x == 0) p[0] += 1;
}
template <int LDS>
__global__ void k_touch(int* p, const float4* src, float4* sink, int n4) {     // every thread reads one 16-byte word of a 4 MB buffer
  __shared__ char pad[LDS > 0 ? LDS : 1];
  if (LDS > 0 && threadIdx.x == 9999) pad[threadIdx.x] = 1;
  const int i = (blockIdx.x * blockDim.x + threadIdx.x) % n4;
  float4 v = src[i];
  if (v.x == 123.456f) sink[i] = v;
  if (blockIdx.x == 0 && threadIdx.x == 0) p[0] += 1;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <typename F> float time_graph(hipStream_t st, int n, F launch) {
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
  for (int i = 0; i < n; ++i) launch();
  hipStreamEndCapture(st, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipGraphLaunch(ge, st); hipStreamSynchronize(st);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a, st);
  for (int r = 0; r < 5; ++r) hipGraphLaunch(ge, st);
  hipEventRecord(b, st); hipStreamSynchronize(st);
  float ms; hipEventElapsedTime(&ms, a, b);
  hipGraphExecDestroy(ge); hipGraphDestroy(g);
  return ms * 1e3f / (5 * n);
}
int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  int* p; CK(hipMalloc(&p, 64)); CK(hipMemset(p, 0, 64));
  float4 *src, *sink; const int n4 = 1 << 18; CK(hipMalloc(&src, n4 * 16)); CK(hipMalloc(&sink, n4 * 16)); CK(hipMemset(src, 0, n4 * 16));
  const int N = 400;
  struct S { int wg, thr; } shapes[] = {{256, 64}, {256, 256}, {256, 512}, {192, 512}, {576, 64}, {768, 64}, {1024, 64}, {2048, 64}, {1024, 256}, {64, 64}, {1, 64}};
  for (auto s : shapes) {
    float e0 = time_graph(st, N, [&] { hipLaunchKernelGGL(k_empty<0>, dim3(s.wg), dim3(s.thr), 0, st, p); });
    float e1 = time_graph(st, N, [&] { hipLaunchKernelGGL(k_empty<131072>, dim3(s.wg), dim3(s.thr), 0, st, p); });
    float t0 = time_graph(st, N, [&] { hipLaunchKernelGGL(k_touch<0>, dim3(s.wg), dim3(s.thr), 0, st, p, src, sink, n4); });
    printf("%5d WG x %4d thr: empty %.2f us | empty + 128 KB LDS %.2f us | one 16 B load per thread %.2f us\n", s.wg, s.thr, e0, e1, t0);
  }
  return 0;
}
