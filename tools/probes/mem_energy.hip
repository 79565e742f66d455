// What a byte costs at each level of the memory hierarchy (package power while a read-only streaming kernel runs from an L2-resident,
// an Infinity-Cache-resident and a DRAM-resident working set): tools/probes/mem_energy.py.  Every workgroup walks `span` bytes starting
// at a workgroup-specific offset of a `total`-byte buffer, `passes` times, with 16-byte loads (8 in flight per lane); the xor-sum keeps
// the loads alive.  xcd_local = 1: workgroup b's region depends only on b % 8 ... so an XCD's workgroups re-read the same `span`
// (L2 hits after the first pass); xcd_local = 0: consecutive workgroups take consecutive regions of the whole buffer.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void stream_read(const u32x4* __restrict__ src, uint64_t total16, uint64_t span16, int passes, int xcd_local,
                                                   unsigned int* __restrict__ sink) {
  const uint64_t region = xcd_local ? (uint64_t)(blockIdx.x & 7) : (uint64_t)blockIdx.x;
  const uint64_t base = (region * span16) % total16;
  u32x4 acc = {0u, 0u, 0u, 0u};
  for (int p = 0; p < passes; ++p) {
    for (uint64_t i = threadIdx.x; i < span16; i += 256 * 8) {
      u32x4 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        uint64_t j = base + i + (uint64_t)k * 256;
        if (j >= total16) j -= total16;
        v[k] = src[j];
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) acc ^= v[k];
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[blockIdx.x] = acc.x;
}

extern "C" void stream_read_launch(const void* src, uint64_t total_bytes, uint64_t span_bytes, int passes, int xcd_local, int blocks,
                                   void* sink, hipStream_t st) {
  hipLaunchKernelGGL(stream_read, dim3(blocks), dim3(256), 0, st, (const u32x4*)src, total_bytes / 16, span_bytes / 16, passes, xcd_local,
                     (unsigned int*)sink);
}
