// Do matrix-core and vector-ALU instructions of DIFFERENT waves on one SIMD overlap?  512-thread workgroups, one per CU: waves
// 0-3 land on SIMDs 0-3, waves 4-7 on the same SIMDs again.  role 0: every wave issues MFMAs; 1: every wave issues v_fma / v_exp;
// 2: waves 0-3 MFMAs, waves 4-7 VALU.  If the two overlap, mode 2 takes max(t_mfma, t_valu) of the one-wave-per-SIMD runs, if
// they serialise it takes the sum.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/probes/coissue.hip -o gpurun_out/coissue.so
#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8v_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

__global__ __launch_bounds__(512) void coissue_kernel(const short* __restrict__ seed, float* __restrict__ out, int iters, int mode,
                                                      int valu_exp) {
  const int wave = threadIdx.x >> 6;
  const bool do_mfma = mode == 0 || (mode == 2 && wave < 4) || (mode == 3 && wave < 4);
  const bool do_valu = mode == 1 || (mode == 2 && wave >= 4) || (mode == 4 && wave >= 4);
  float s = 0.f;
  if (do_mfma) {
    s16x8_t a, b;
    for (int e = 0; e < 8; ++e) { a[e] = seed[(threadIdx.x * 8 + e) & 4095]; b[e] = seed[(threadIdx.x * 8 + e + 1777) & 4095]; }
    f32x16_t acc[4];
    for (int i = 0; i < 4; ++i)
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8v_t, a), __builtin_bit_cast(bf16x8v_t, b), acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i)
      for (int r = 0; r < 16; ++r) s += acc[i][r];
  } else if (do_valu) {
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = (float)seed[(threadIdx.x + i * 64) & 4095] * 1e-3f;
    const float c = (float)seed[7] * 1e-5f + 0.999f, d = 1e-3f;
    // per iteration: 32 VALU instructions (16 independent chains, 2 each) = the issue time of 4 32x32x16 MFMAs (4 x 32 clocks)
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        x[i] = __builtin_fmaf(x[i], c, d);
        if (valu_exp) x[i] = __builtin_amdgcn_exp2f(x[i] - 1.f); else x[i] = __builtin_fmaf(x[i], c, -d);
      }
    }
    for (int i = 0; i < 16; ++i) s += x[i];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

extern "C" int coissue_launch(const void* seed, void* out, int blocks, int iters, int mode, int valu_exp, hipStream_t st) {
  hipLaunchKernelGGL(coissue_kernel, dim3(blocks), dim3(512), 0, st, (const short*)seed, (float*)out, iters, mode, valu_exp);
  return (int)hipGetLastError();
}
