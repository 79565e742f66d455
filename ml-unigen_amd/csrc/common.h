// Shared device/host helpers for the UniGen gfx950 kernels.
// Everything here is CDNA4-only (wave64, MFMA); there is no other backend.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

// ---------------------------------------------------------------- error plumbing
// No exceptions / aborts cross the C-ABI: entry points return 0 or a negative error class and
// leave a message readable through ug_last_error().
#define UG_OK 0
#define UG_ERR_ARG (-1)     // bad shape / alignment / dtype
#define UG_ERR_LAUNCH (-2)  // HIP launch or runtime failure
#define UG_ERR_STATE (-3)   // handle / workspace misuse

void ug_set_error(const char* fmt, ...);

#define UG_REQUIRE(cond, ...)                  \
  do {                                         \
    if (!(cond)) {                             \
      ug_set_error(__VA_ARGS__);               \
      return UG_ERR_ARG;                       \
    }                                          \
  } while (0)

#define UG_CHECK_LAUNCH(name)                                                     \
  do {                                                                            \
    hipError_t e__ = hipGetLastError();                                           \
    if (e__ != hipSuccess) {                                                      \
      ug_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));        \
      return UG_ERR_LAUNCH;                                                       \
    }                                                                             \
  } while (0)

#define UG_HIP(call)                                                              \
  do {                                                                            \
    hipError_t e__ = (call);                                                      \
    if (e__ != hipSuccess) {                                                      \
      ug_set_error("%s failed: %s", #call, hipGetErrorString(e__));               \
      return UG_ERR_LAUNCH;                                                       \
    }                                                                             \
  } while (0)

// ---------------------------------------------------------------- handle (include/unigen_hip.h: ug_create / ug_destroy)
#define UG_HANDLE_WS_SLOTS 768                      // x 256 KiB of fp32 partials
struct ug_handle {
  float* tail_ws;                                   // [UG_HANDLE_WS_SLOTS][256][256] fp32, owned
};

static inline bool ug_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ---------------------------------------------------------------- vector types
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;   // 8 bf16 = 4 VGPRs (MFMA A/B operand)
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;
typedef __attribute__((ext_vector_type(2))) short bf16x2_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;    // 16x16 MFMA accumulator
typedef __attribute__((ext_vector_type(16))) float f32x16_t;  // 32x32 MFMA accumulator
typedef unsigned short bf16_t;                                 // raw bf16 bits

#define UG_WAVE 64

// ---------------------------------------------------------------- bf16 <-> f32
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even, NaN stays NaN (same rounding torch uses for .to(bfloat16)): gfx950's packed hardware
// conversion, one VALU instruction per two values (the integer sequence it replaces cost ~6 per value and was a
// visible share of the attention kernels' VALU time)
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  typedef __bf16 hwbf16x2_t __attribute__((ext_vector_type(2)));
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, hwbf16x2_t));    // v_cvt_pk_bf16_f32
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack_bf2(f, f) & 0xffffu); }

// ---------------------------------------------------------------- SiLU of the training forward (one definition for every kernel)
// silu(g) = g / (1 + exp(-g)) with the quotient as rcp + one residual correction (6 VALU operations; the IEEE division sequence
// -- div_scale x 2, rcp, five fma, div_fmas, div_fixup -- was 34 of the fused gate_up epilogue's 60 extra us per launch).  The
// divisor is in [1, inf): no scaling needed; exp(-g) = inf (g < -88.7) gives the signed zero the exact quotient rounds to.
// swiglu_fwd_kernel and the GEMM epilogues share it, so the fused and the two-launch forms stay bit-identical to each other;
// against the fp32 reference the quotient is within 1 ulp before the bf16 rounding that follows it everywhere.
__device__ __forceinline__ float silu_train(float g) {
  const float d = 1.f + __expf(-g);
  const float y = __builtin_amdgcn_rcpf(d);
  const float q = g * y;
  const float r = __builtin_fmaf(-d, q, g);
  const float v = __builtin_fmaf(r, y, q);
  return (d > 3.0e38f) ? g * 0.f : v;
}

// One element of the SwiGLU backward (autograd of act = bf16(bf16(silu(gate)) * up) under bf16 autocast, Qwen2MLP.forward,
// modeling_qwen2.py:46-48): shared by swiglu_bwd_kernel and the down-projection dgrad's epilogue (EPI_SWIGLU_BWD) with contraction off,
// so the fused and the two-launch forms are bit-identical to each other.  sigmoid = rcp + one Newton step (divisor in [1, inf)).
__device__ __forceinline__ void swiglu_bwd_elem(float gf, float uf, float df, float& dgate, float& dup) {
#pragma clang fp contract(off)
  const float d = 1.f + __expf(-gf);
  const float y0 = __builtin_amdgcn_rcpf(d);
  const float y1 = __builtin_fmaf(y0, __builtin_fmaf(-d, y0, 1.f), y0);
  const float sg = (d > 3.0e38f) ? 0.f : y1;
  const float s = bf2f(f2bf(gf * sg));                    // bf16 silu output saved by autograd
  const float dsilu = bf2f(f2bf(df * uf));                // grad wrt silu output (bf16 mul backward)
  dup = df * s;
  dgate = dsilu * (sg * (1.f + gf * (1.f - sg)));
}

// ---------------------------------------------------------------- wave reductions (64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Block-wide sum for blocks of NW waves; every thread gets the result. `red` is NW floats of LDS.
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if (l == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < NW; ++i) t += red[i];
  return t;
}
template <int NW>
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if (l == 0) red[w] = v;
  __syncthreads();
  float t = red[0];
#pragma unroll
  for (int i = 1; i < NW; ++i) t = fmaxf(t, red[i]);
  return t;
}

// XCD-aware, bijective remap of a linear workgroup id: consecutive ids are dispatched round-robin
// over the 8 XCDs (private L2 each); this gives every XCD one contiguous chunk of the tile grid.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}
