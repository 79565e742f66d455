#include <hip/hip_runtime.h>
#include <cstdio>
// mode 0: src[global id]  1: src[threadIdx.x] (same 1 KB for every WG)  2: src[global id] non-temporal  3: scalar (uniform) load per WG
template <int MODE>
__global__ void k_touch(int* p, const float4* src, float4* sink, int n4) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (MODE == 1) i = threadIdx.x;
  if (MODE == 3) i = blockIdx.x * 64;
  i &= (n4 - 1);
  float4 v;
  if (MODE == 2) { typedef float f4 __attribute__((ext_vector_type(4))); f4 t = __builtin_nontemporal_load(reinterpret_cast<const f4*>(src) + i); v = make_float4(t.x, t.y, t.z, t.w); }
  else v = src[i];
  if (v.x == 123.456f) sink[i] = v;
  if (blockIdx.x == 0 && threadIdx.x == 0) p[0] += 1;
}
template <typename F> float time_graph(hipStream_t st, int n, F launch) {
  hipGraph_t g; hipGraphExec_t ge;
  (void)hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
  for (int i = 0; i < n; ++i) launch(i);
  (void)hipStreamEndCapture(st, &g);
  (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  (void)hipGraphLaunch(ge, st); (void)hipStreamSynchronize(st);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  (void)hipEventRecord(a, st);
  for (int r = 0; r < 5; ++r) (void)hipGraphLaunch(ge, st);
  (void)hipEventRecord(b, st); (void)hipStreamSynchronize(st);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  return ms * 1e3f / (5 * n);
}
int main() {
  hipStream_t st; (void)hipStreamCreate(&st);
  int* p; (void)hipMalloc(&p, 64); (void)hipMemset(p, 0, 64);
  float4 *src, *sink; const int n4 = 1 << 22; (void)hipMalloc(&src, (size_t)n4 * 16); (void)hipMalloc(&sink, (size_t)n4 * 16); (void)hipMemset(src, 0, (size_t)n4 * 16);
  const int N = 200;
  int wgs[] = {1, 16, 64, 256, 1024};
  for (int wg : wgs) {
    float t0 = time_graph(st, N, [&](int) { hipLaunchKernelGGL(k_touch<0>, dim3(wg), dim3(64), 0, st, p, src, sink, n4); });
    float t1 = time_graph(st, N, [&](int) { hipLaunchKernelGGL(k_touch<1>, dim3(wg), dim3(64), 0, st, p, src, sink, n4); });
    float t2 = time_graph(st, N, [&](int) { hipLaunchKernelGGL(k_touch<2>, dim3(wg), dim3(64), 0, st, p, src, sink, n4); });
    float t3 = time_graph(st, N, [&](int) { hipLaunchKernelGGL(k_touch<3>, dim3(wg), dim3(64), 0, st, p, src, sink, n4); });
    // a different 1 MB window of the buffer per launch (nothing re-read within a replay)
    float t4 = time_graph(st, N, [&](int i) { hipLaunchKernelGGL(k_touch<0>, dim3(wg), dim3(64), 0, st, p, src + (size_t)(i % 60) * 65536, sink, 65536); });
    printf("%5d WG x 64: src[gid] %.2f | src[tid] %.2f | nt %.2f | one line per WG %.2f | fresh window per launch %.2f us\n", wg, t0, t1, t2, t3, t4);
  }
  // eager (no graph) launches
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int wg : wgs) {
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_touch<0>, dim3(wg), dim3(64), 0, st, p, src, sink, n4);
    (void)hipEventRecord(a, st);
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_touch<0>, dim3(wg), dim3(64), 0, st, p, src, sink, n4);
    (void)hipEventRecord(b, st); (void)hipStreamSynchronize(st);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    printf("%5d WG x 64 eager: %.2f us per launch\n", wg, ms * 1e3f / 200);
  }
  return 0;
}
