// CU-contention model for the data-parallel exchange on ONE GPU: a copy kernel of exactly G workgroups
// (what an RCCL ring all-reduce holds: one workgroup per channel) that moves `bytes` through HBM (read once, write once) at
// whatever rate G workgroups reach -- the stand-in for a bucket's reduce-scatter + all-gather on the side stream while the
// backward GEMMs assume every CU.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/probes/occupy.hip -o tools/probes/_build/occupy.so
#include <hip/hip_runtime.h>
#include <stdint.h>
__global__ __launch_bounds__(256) void occupy_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, int64_t n16, int passes) {
  for (int p = 0; p < passes; ++p)
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x) {
      uint4 v = src[i];
      v.x += p;
      dst[i] = v;
    }
}
extern "C" int occupy_launch(const void* src, void* dst, int64_t bytes, int groups, int passes, hipStream_t st) {
  hipLaunchKernelGGL(occupy_copy, dim3(groups), dim3(256), 0, st, (const uint4*)src, (uint4*)dst, bytes / 16, passes);
  return (int)hipGetLastError();
}
