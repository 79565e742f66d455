// LDS read bandwidth probe for gfx950: every wave streams conflict-free ds_read_b128 fragments (the bf16 GEMM's
// fragment pattern: 16 rows x 64 B, XOR-swizzled) out of a 64 KB LDS tile, optionally with MFMAs consuming them.
#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

typedef __attribute__((ext_vector_type(4))) short s16x4_t;
// TR: fragments through two ds_read_tr16_b64 (the k-major operand path) instead of one ds_read_b128
#ifndef SWZ_OLD
#define SWZ_OLD 0     // 1: round 1's row-quad map (0,1,2,3), 2-way conflicts on the real ds_read_b128 lane groups
#endif
template <int MFMA_PER_READ, bool TR = false>
__global__ __launch_bounds__(512) void lds_read_kernel(float* __restrict__ out, int iters) {
  __shared__ __attribute__((aligned(16))) char tile[65536];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 65536 / 4; i += blockDim.x) reinterpret_cast<int*>(tile)[i] = i * 2654435761u;
  __syncthreads();
  const int row = lane & 15, g = lane >> 4;
  f32x4_t acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  bf16x8_t keep = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int r = ((wave * 16 + u) * 16 + row + it) & 1023;                 // 1024 rows x 64 B
      bf16x8_t f;
      if constexpr (!TR) {
        f = *reinterpret_cast<const bf16x8_t*>(tile + r * 64 + ((g ^ ((SWZ_OLD ? (r >> 2) : (0 - (r >> 2))) & 3)) << 4));
      } else {
        const char* p0 = tile + (((wave * 16 + u) * 1024 + it * 64) & 65535 & ~1023) + lane * 8;      // 512 contiguous bytes per read
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p0);
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p0 + 512));
        f = bf16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
      if constexpr (MFMA_PER_READ == 0) {
        keep ^= f;
      } else {
#pragma unroll
        for (int m = 0; m < MFMA_PER_READ; ++m) acc[(u + m) & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f, f, acc[(u + m) & 3], 0, 0, 0);
      }
    }
  }
  float s = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
  for (int k = 0; k < 8; ++k) s += (float)keep[k];
  if (s == 1.2345f) out[blockIdx.x * blockDim.x + tid] = s;
}

extern "C" void lds_read_launch(float* out, int blocks, int threads, int iters, int mfma_per_read, hipStream_t st) {
  if (mfma_per_read >= 100) {
    switch (mfma_per_read - 100) {
      case 0: hipLaunchKernelGGL((lds_read_kernel<0, true>), dim3(blocks), dim3(threads), 0, st, out, iters); break;
      case 2: hipLaunchKernelGGL((lds_read_kernel<2, true>), dim3(blocks), dim3(threads), 0, st, out, iters); break;
      default: hipLaunchKernelGGL((lds_read_kernel<3, true>), dim3(blocks), dim3(threads), 0, st, out, iters); break;
    }
    return;
  }
  switch (mfma_per_read) {
    case 0: hipLaunchKernelGGL(lds_read_kernel<0>, dim3(blocks), dim3(threads), 0, st, out, iters); break;
    case 1: hipLaunchKernelGGL(lds_read_kernel<1>, dim3(blocks), dim3(threads), 0, st, out, iters); break;
    case 2: hipLaunchKernelGGL(lds_read_kernel<2>, dim3(blocks), dim3(threads), 0, st, out, iters); break;
    case 3: hipLaunchKernelGGL(lds_read_kernel<3>, dim3(blocks), dim3(threads), 0, st, out, iters); break;
    default: hipLaunchKernelGGL(lds_read_kernel<4>, dim3(blocks), dim3(threads), 0, st, out, iters); break;
  }
}
