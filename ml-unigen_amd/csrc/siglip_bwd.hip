// Backward pieces of the SigLIP vision tower for the UNFROZEN case (reference models/unigen.py:111 `freeze=False`,
// training/train_w_clip_vit.py:282,311-312: 'mm_vision_tower' among the tunable parts).  No shipped YAML trains the tower, so
// these kernels favour exactness over speed: fp32 throughout, the contractions of the backward run on the exact fp32 MFMA GEMMs
// (ug_gemm_f32 / ug_gemm_f32_nested, conv_f32.hip) and only the element-wise / row-wise derivatives live here.
//   LayerNorm backward          nn.LayerNorm(eps 1e-6), siglip_encoder.py:267-269,296-309
//   gelu_pytorch_tanh fwd / bwd SigLipMLP, :256-259 (training keeps the pre-activation, so the forward is a separate pass)
//   softmax backward            SigLipAttention, :219-229 (fp32 softmax over keys)
//   column sums                 bias gradients of the six Linear layers of a block, patch / position embedding
#include "common.h"
#include "unigen_hip.h"

namespace {

// one wave per row: dx = rstd * (dxhat - mean(dxhat) - xhat * mean(dxhat * xhat)), dxhat = dy * gamma; the statistics are
// recomputed from x (two passes, like the forward kernel).  dgamma += sum_rows dy * xhat, dbeta += sum_rows dy: per-block
// partials in LDS, one atomic per column per block.  dx_out = dres_in + dx when dres_in is given (the residual branch).
__global__ __launch_bounds__(256) void layernorm_bwd_f32_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                const float* __restrict__ gamma, const float* __restrict__ dres_in,
                                                                float* __restrict__ dx, float* __restrict__ dgamma,
                                                                float* __restrict__ dbeta, int rows, int cols, float eps,
                                                                int rows_per_block) {
  extern __shared__ float part[];                      // [2][cols]: dgamma | dbeta partials of this block
  for (int c = threadIdx.x; c < 2 * cols; c += blockDim.x) part[c] = 0.f;
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  for (int row = r0 + wave; row < r1; row += 4) {
    const float* xr = x + (int64_t)row * cols;
    const float* dyr = dy + (int64_t)row * cols;
    float s = 0.f;
    for (int c = lane; c < cols; c += 64) s += xr[c];
    const float mean = wave_sum(s) / (float)cols;
    float v = 0.f;
    for (int c = lane; c < cols; c += 64) { const float d = xr[c] - mean; v += d * d; }
    const float rstd = rsqrtf(wave_sum(v) / (float)cols + eps);
    float a = 0.f, b = 0.f;
    for (int c = lane; c < cols; c += 64) {
      const float xh = (xr[c] - mean) * rstd, g = dyr[c] * gamma[c];
      a += g; b += g * xh;
    }
    a = wave_sum(a) / (float)cols;
    b = wave_sum(b) / (float)cols;
    for (int c = lane; c < cols; c += 64) {
      const float xh = (xr[c] - mean) * rstd, d = dyr[c];
      float o = rstd * (d * gamma[c] - a - xh * b);
      if (dres_in) o += dres_in[(int64_t)row * cols + c];
      dx[(int64_t)row * cols + c] = o;
      atomicAdd(&part[c], d * xh);                      // LDS atomics: four waves of a block meet in the same columns
      atomicAdd(&part[cols + c], d);
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < cols; c += blockDim.x) {
    atomicAdd(dgamma + c, part[c]);
    atomicAdd(dbeta + c, part[cols + c]);
  }
}

__device__ __forceinline__ float gelu_tanh_f(float x) {
  const float k = 0.7978845608028654f;                // sqrt(2 / pi)
  return 0.5f * x * (1.f + tanhf(k * (x + 0.044715f * x * x * x)));
}
// mode 0: out = gelu(pre);  mode 1: out = dy * gelu'(pre)
__global__ __launch_bounds__(256) void gelu_tanh_f32_kernel(const float* __restrict__ pre, const float* __restrict__ dy,
                                                            float* __restrict__ out, int64_t n, int mode) {
  const float k = 0.7978845608028654f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float x = pre[i];
    if (mode == 0) { out[i] = gelu_tanh_f(x); continue; }
    const float u = k * (x + 0.044715f * x * x * x), t = tanhf(u);
    const float du = k * (1.f + 3.f * 0.044715f * x * x);
    out[i] = dy[i] * (0.5f * (1.f + t) + 0.5f * x * (1.f - t * t) * du);
  }
}

// in place on dP (one wave per row): dS = scale * P * (dP - sum_j dP_j P_j) over the first `cols` columns; the padding columns
// [cols, ld) are set to zero (the contraction that follows runs over a 16-byte-aligned width)
__global__ __launch_bounds__(256) void softmax_bwd_rows_f32_kernel(const float* __restrict__ P, float* __restrict__ dP, int64_t rows,
                                                                   int cols, int64_t ld, float scale) {
  const int64_t row = blockIdx.x * 4LL + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* p = P + row * ld;
  float* d = dP + row * ld;
  float s = 0.f;
  for (int c = lane; c < cols; c += 64) s += d[c] * p[c];
  s = wave_sum(s);
  for (int c = lane; c < (int)ld; c += 64) d[c] = c < cols ? scale * p[c] * (d[c] - s) : 0.f;
}

// out[c] (+)= sum_r x[r, c]: each block sums a stripe of rows into registers, one atomic per column per block
__global__ __launch_bounds__(256) void colsum_f32_kernel(const float* __restrict__ x, int64_t ld, float* __restrict__ out,
                                                         int64_t rows, int cols, int rows_per_block) {
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  float s = 0.f;
  for (int64_t r = r0; r < r1; ++r) s += x[r * ld + c];
  atomicAdd(out + c, s);
}

// batched 2-D transpose through a 32 x 33 LDS tile: out[z][c][r] = in[z][r][c]; output rows are ld_out long, columns
// [rows, ld_out) are set to zero (the transposed tensor is the A operand of an fp32 GEMM whose contraction runs over a
// 16-byte-aligned width)
__global__ __launch_bounds__(256) void transpose_f32_kernel(const float* __restrict__ in, int64_t ld_in, int64_t stride_in,
                                                            float* __restrict__ out, int64_t ld_out, int64_t stride_out, int rows,
                                                            int cols) {
  __shared__ float tile[32][33];
  const float* src = in + blockIdx.z * stride_in;
  float* dst = out + blockIdx.z * stride_out;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8 threads
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < rows && c < cols) ? src[(int64_t)r * ld_in + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + tx;                              // output row c, output column r
    if (c < cols && r < ld_out) dst[(int64_t)c * ld_out + r] = tile[tx][i];
  }
}

}  // namespace

extern "C" int ug_layernorm_bwd_f32(const float* dy, const float* x, const float* gamma, const float* dres_in, float* dx,
                                    float* dgamma, float* dbeta, int64_t rows, int64_t cols, float eps, hipStream_t st) {
  UG_REQUIRE(rows > 0 && cols > 0 && cols <= 8192 && dy && x && gamma && dx && dgamma && dbeta,
             "ug_layernorm_bwd_f32: need rows > 0, 0 < cols <= 8192 and non-null buffers");
  const int rpb = 32;
  hipLaunchKernelGGL(layernorm_bwd_f32_kernel, dim3((unsigned)((rows + rpb - 1) / rpb)), dim3(256), 2 * cols * sizeof(float), st, dy, x,
                     gamma, dres_in, dx, dgamma, dbeta, (int)rows, (int)cols, eps, rpb);
  UG_CHECK_LAUNCH("ug_layernorm_bwd_f32");
  return UG_OK;
}

extern "C" int ug_gelu_tanh_f32(const float* pre, const float* dy_or_null, float* out, int64_t n, hipStream_t st) {
  UG_REQUIRE(n > 0 && pre && out, "ug_gelu_tanh_f32: empty");
  int64_t g = (n + 255) / 256; if (g > 8192) g = 8192;
  hipLaunchKernelGGL(gelu_tanh_f32_kernel, dim3((unsigned)g), dim3(256), 0, st, pre, dy_or_null, out, n, dy_or_null ? 1 : 0);
  UG_CHECK_LAUNCH("ug_gelu_tanh_f32");
  return UG_OK;
}

extern "C" int ug_softmax_bwd_rows_f32(const float* P, float* dP, int64_t rows, int64_t cols, int64_t ld, float scale, hipStream_t st) {
  UG_REQUIRE(rows > 0 && cols > 0 && ld >= cols && P && dP, "ug_softmax_bwd_rows_f32: bad args");
  hipLaunchKernelGGL(softmax_bwd_rows_f32_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, P, dP, rows, (int)cols, ld, scale);
  UG_CHECK_LAUNCH("ug_softmax_bwd_rows_f32");
  return UG_OK;
}

extern "C" int ug_colsum_f32(const float* x, int64_t ld, float* out, int64_t rows, int64_t cols, hipStream_t st) {
  UG_REQUIRE(rows > 0 && cols > 0 && ld >= cols && x && out, "ug_colsum_f32: bad args");
  const int rpb = 64;
  hipLaunchKernelGGL(colsum_f32_kernel, dim3((unsigned)((cols + 255) / 256), (unsigned)((rows + rpb - 1) / rpb)), dim3(256), 0, st, x, ld,
                     out, rows, (int)cols, rpb);
  UG_CHECK_LAUNCH("ug_colsum_f32");
  return UG_OK;
}

extern "C" int ug_transpose_f32(const float* in, int64_t ld_in, int64_t stride_in, float* out, int64_t ld_out, int64_t stride_out,
                                int64_t rows, int64_t cols, int64_t batch, hipStream_t st) {
  UG_REQUIRE(rows > 0 && cols > 0 && batch > 0 && batch < 65536 && ld_in >= cols && ld_out >= rows && in && out,
             "ug_transpose_f32: bad args");
  dim3 grid((unsigned)((cols + 31) / 32), (unsigned)((ld_out + 31) / 32), (unsigned)batch);
  hipLaunchKernelGGL(transpose_f32_kernel, grid, dim3(256), 0, st, in, ld_in, stride_in, out, ld_out, stride_out, (int)rows, (int)cols);
  UG_CHECK_LAUNCH("ug_transpose_f32");
  return UG_OK;
}
