// Achievable bf16 MFMA rate of the whole chip with nothing but MFMAs in flight (no memory traffic):
// the practical ceiling the GEMM kernels should be judged against next to the 2.5 PFLOP/s datasheet figure.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/probes/mfma_peak.hip -o gpurun_out/mfma_peak.so
#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

template <int NACC>
__global__ __launch_bounds__(256) void mfma_burn(const short* __restrict__ seed, float* __restrict__ out, int iters) {
  bf16x8_t a, b;
  for (int e = 0; e < 8; ++e) { a[e] = seed[(threadIdx.x * 8 + e) & 4095]; b[e] = seed[(threadIdx.x * 8 + e + 1777) & 4095]; }
  f32x4_t acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

typedef __attribute__((ext_vector_type(16))) float f32x16_t;
// fp32-input matrix cores (the tokenizer's convolutions): v_mfma_f32_32x32x2_f32
template <int NACC>
__global__ __launch_bounds__(256) void mfma_burn_f32(const short* __restrict__ seed, float* __restrict__ out, int iters) {
  const float a = (float)seed[threadIdx.x & 4095] * 1e-4f, b = (float)seed[(threadIdx.x + 977) & 4095] * 1e-4f;
  f32x16_t acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}


// v_mfma_f32_32x32x16_{bf16,f16}: 16 accumulator registers per block, 32 issue cycles per SIMD; one wave per SIMD is
// documented to issue them back to back (MI355X_MICROARCH.md: 2 495 TF/s micro-benchmark).
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8v_t;
template <int NACC, bool F16>
__global__ __launch_bounds__(256) void mfma_burn32(const short* __restrict__ seed, float* __restrict__ out, int iters) {
  bf16x8_t a, b;
  for (int e = 0; e < 8; ++e) { a[e] = seed[(threadIdx.x * 8 + e) & 4095]; b[e] = seed[(threadIdx.x * 8 + e + 1777) & 4095]; }
  f32x16_t acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      if constexpr (F16) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), acc[i], 0, 0, 0);
      else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8v_t, a), __builtin_bit_cast(bf16x8v_t, b), acc[i], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

extern "C" int mfma_burn32_launch(const void* seed, void* out, int blocks, int threads, int iters, int nacc, int f16, hipStream_t st) {
  if (f16) {
    if (nacc == 16) hipLaunchKernelGGL((mfma_burn32<16, true>), dim3(blocks), dim3(threads), 0, st, (const short*)seed, (float*)out, iters);
    else hipLaunchKernelGGL((mfma_burn32<4, true>), dim3(blocks), dim3(threads), 0, st, (const short*)seed, (float*)out, iters);
  } else {
    if (nacc == 16) hipLaunchKernelGGL((mfma_burn32<16, false>), dim3(blocks), dim3(threads), 0, st, (const short*)seed, (float*)out, iters);
    else hipLaunchKernelGGL((mfma_burn32<4, false>), dim3(blocks), dim3(threads), 0, st, (const short*)seed, (float*)out, iters);
  }
  return (int)hipGetLastError();
}

extern "C" int mfma_burn_f32_launch(const void* seed, void* out, int blocks, int threads, int iters, hipStream_t st) {
  hipLaunchKernelGGL(mfma_burn_f32<4>, dim3(blocks), dim3(threads), 0, st, (const short*)seed, (float*)out, iters);
  return (int)hipGetLastError();
}

extern "C" int mfma_burn_launch(const void* seed, void* out, int blocks, int threads, int iters, int nacc, hipStream_t st) {
  if (nacc == 64) hipLaunchKernelGGL(mfma_burn<64>, dim3(blocks), dim3(threads), 0, st, (const short*)seed, (float*)out, iters);
  else hipLaunchKernelGGL(mfma_burn<16>, dim3(blocks), dim3(threads), 0, st, (const short*)seed, (float*)out, iters);
  return (int)hipGetLastError();
}
