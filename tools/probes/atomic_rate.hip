// What dQ accumulated by fp32 atomics would cost in a ONE-kernel attention backward.  Build: hipcc
// --offload-arch=gfx950 -O3 -munsafe-fp-atomics atomic_rate.hip -o _build/atomic_rate
// One 256-thread workgroup per (query tile of 64 rows, pass): adds a 64 x 128 fp32 tile (32 KB, row-contiguous: every atomic
// instruction covers 64 lanes x 4 consecutive bytes x 4 rows ... here one float per lane, 256 consecutive bytes per wave) into
// dQ [tiles][64][128]; `passes` workgroups hit the same tile (the key tiles that see it), spread over the launch.
// 16 x 771 tokens x 12 heads = 2 316 tiles of 64 rows = 75.9 MB of fp32; 13 key tiles (7 on average under a causal mask).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int SCOPE>
__global__ __launch_bounds__(256) void add_tiles(float* __restrict__ dq, int ntiles, int same_xcd) {
  // workgroup -> (tile, pass).  same_xcd: every pass of a tile on one XCD (b % 8 decides); else consecutive passes on consecutive XCDs
  int tile, pass;
  const int npass = gridDim.x / ntiles;
  if (same_xcd) { const int x = blockIdx.x & 7, j = blockIdx.x >> 3; pass = j % npass; tile = (j / npass) * 8 + x; }
  else { tile = blockIdx.x / npass; pass = blockIdx.x % npass; }
  if (tile >= ntiles) return;
  float* t = dq + (size_t)tile * 64 * 128;
  const float v = 1.0f + pass;
#pragma unroll 8
  for (int i = threadIdx.x; i < 64 * 128; i += 256) {
    if (SCOPE == 0) atomicAdd(t + i, v);                                                                    // device scope (what a product may use)
    else __hip_atomic_fetch_add(t + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);                  // resolves in the XCD's L2
  }
}
__global__ void plain_store(float* dq, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dq[i] = 1.f;
}
int main() {
  const int ntiles = 2320;                 // multiple of 8
  const size_t n = (size_t)ntiles * 64 * 128;
  float* dq; CK(hipMalloc(&dq, n * 4)); CK(hipMemset(dq, 0, n * 4));
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int npass : {1, 7, 13}) {
    for (int same = 0; same < 2; ++same)
      for (int scope = 0; scope < 2; ++scope) {
        float best = 1e9;
        for (int rep = 0; rep < 4; ++rep) {
          hipEventRecord(a);
          if (scope == 0) hipLaunchKernelGGL(add_tiles<0>, dim3(ntiles * npass), dim3(256), 0, 0, dq, ntiles, same);
          else hipLaunchKernelGGL(add_tiles<1>, dim3(ntiles * npass), dim3(256), 0, 0, dq, ntiles, same);
          hipEventRecord(b); CK(hipDeviceSynchronize());
          float ms; hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best;
        }
        printf("passes %2d  %s  %s-scope atomics: %8.1f us  (%.2f TB/s of fp32 added)\n", npass, same ? "tile's passes on ONE XCD " : "passes on different XCDs",
               scope ? "workgroup" : "device   ", best * 1e3, n * 4.0 * npass / (best * 1e-3) / 1e12);
      }
  }
  float best = 1e9;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(a); hipLaunchKernelGGL(plain_store, dim3(2048), dim3(256), 0, 0, dq, n); hipEventRecord(b); CK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best;
  }
  printf("plain stores of the same %.1f MB: %.1f us\n", n * 4 / 1e6, best * 1e3);
  return 0;
}
