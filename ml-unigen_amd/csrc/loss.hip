// Cross-entropy over the 159 867-entry tied vocabulary (reference: the three F.cross_entropy calls
// of UniGen.forward, models/unigen.py:310-338; DPO's get_batch_logps, training/train_dpo.py:51-90).
// Logits arrive as bf16 rows [R, ld] straight from the lm_head GEMM; statistics are fp32 like the
// reference's autocast-promoted F.cross_entropy.  One workgroup per row, online max/sum in a single
// pass (the 320 KB row stays in L2 for the gradient pass), gradient written in place as bf16.
#include "common.h"
#include "unigen_hip.h"

namespace {

__global__ __launch_bounds__(256) void ce_fwd_kernel(const bf16_t* __restrict__ logits, int64_t ld, int V,
                                                     const int64_t* __restrict__ labels, int64_t ignore_index,
                                                     float* __restrict__ lse_out, float* __restrict__ loss_row,
                                                     float* __restrict__ logp_label) {
  __shared__ float red[4];
  const int row = blockIdx.x;
  const bf16_t* x = logits + (int64_t)row * ld;
  const int nv = V >> 3;
  float m = -INFINITY, s = 0.f;
  for (int i = threadIdx.x; i < nv; i += 256) {
    const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(x + i * 8);
    float f[8], mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < 8; ++k) { f[k] = bf2f((bf16_t)v[k]); mx = fmaxf(mx, f[k]); }
    const float mn = fmaxf(m, mx);
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += __expf(f[k] - mn);
    s = s * __expf(m - mn) + acc;
    m = mn;
  }
  for (int c = (nv << 3) + threadIdx.x; c < V; c += 256) {   // ragged tail (V % 8)
    const float f = bf2f(x[c]);
    const float mn = fmaxf(m, f);
    s = s * __expf(m - mn) + __expf(f - mn);
    m = mn;
  }
  const float M = block_max<4>(m, red);
  const float part = (m == -INFINITY) ? 0.f : s * __expf(m - M);
  const float S = block_sum<4>(part, red);
  if (threadIdx.x == 0) {
    const float lse = M + logf(S);
    lse_out[row] = lse;
    const int64_t lab = labels ? labels[row] : ignore_index;
    float lr = 0.f, lp = 0.f;
    if (lab != ignore_index && lab >= 0 && lab < V) { lp = bf2f(x[lab]) - lse; lr = -lp; }
    if (loss_row) loss_row[row] = lr;
    if (logp_label) logp_label[row] = lp;
  }
}

// loss = sum(loss_row over valid rows) / count ;  out[0] = loss, out[1] = count
__global__ __launch_bounds__(256) void ce_reduce_kernel(const float* __restrict__ loss_row,
                                                        const int64_t* __restrict__ labels, int64_t ignore_index,
                                                        int R, float* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.f, c = 0.f;
  for (int i = threadIdx.x; i < R; i += 256) {
    if (labels[i] != ignore_index) { s += loss_row[i]; c += 1.f; }
  }
  const float S = block_sum<4>(s, red);
  const float C = block_sum<4>(c, red);
  if (threadIdx.x == 0) { out[0] = S / C; out[1] = C; }
}

// dlogits = (softmax - onehot) * (gscale / count) for valid rows, 0 otherwise; pad columns zeroed
__global__ __launch_bounds__(256) void ce_bwd_kernel(bf16_t* __restrict__ logits, int64_t ld, int V,
                                                     const int64_t* __restrict__ labels, int64_t ignore_index,
                                                     const float* __restrict__ lse, const float* __restrict__ loss_cnt,
                                                     const float* __restrict__ gscale, const float* __restrict__ row_scale) {
  const int row = blockIdx.x;
  bf16_t* x = logits + (int64_t)row * ld;
  const int64_t lab = labels[row];
  const bool valid = (lab != ignore_index);
  // mean-CE gradient (*gscale / count) or, with row_scale, an arbitrary upstream gradient per row (log-prob sums of DPO)
  const float sc = !valid ? 0.f : row_scale ? row_scale[row] : (gscale ? *gscale : 1.f) / loss_cnt[1];
  const float l = lse[row];
  const int nvp = (int)(ld >> 3);
  for (int i = threadIdx.x; i < nvp; i += 256) {
    bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(x + i * 8);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int c = i * 8 + k;
      float g = 0.f;
      if (valid && c < V) g = (__expf(bf2f((bf16_t)v[k]) - l) - (c == lab ? 1.f : 0.f)) * sc;
      v[k] = (short)f2bf(g);
    }
    *reinterpret_cast<bf16x8_t*>(x + i * 8) = v;
  }
}

}  // namespace

extern "C" int ug_ce_fwd(const void* logits, int64_t ld, int64_t R, int64_t V, const int64_t* labels,
                         int64_t ignore_index, float* lse, float* loss_row, float* logp_label, float* loss_and_count,
                         hipStream_t st) {
  UG_REQUIRE(R > 0 && V > 0 && ld >= V && ld % 8 == 0, "ug_ce_fwd: need ld>=V and ld%%8==0 (ld=%ld V=%ld)", (long)ld, (long)V);
  UG_REQUIRE(ug_aligned16(logits), "ug_ce_fwd: logits must be 16B aligned");
  hipLaunchKernelGGL(ce_fwd_kernel, dim3((unsigned)R), dim3(256), 0, st, (const bf16_t*)logits, ld, (int)V, labels,
                     ignore_index, lse, loss_row, logp_label);
  UG_CHECK_LAUNCH("ug_ce_fwd");
  if (loss_and_count) {
    UG_REQUIRE(labels && loss_row, "ug_ce_fwd: loss reduction needs labels and loss_row");
    hipLaunchKernelGGL(ce_reduce_kernel, dim3(1), dim3(256), 0, st, loss_row, labels, ignore_index, (int)R, loss_and_count);
    UG_CHECK_LAUNCH("ug_ce_fwd(reduce)");
  }
  return UG_OK;
}

extern "C" int ug_ce_bwd(void* logits_inout, int64_t ld, int64_t R, int64_t V, const int64_t* labels,
                         int64_t ignore_index, const float* lse, const float* loss_and_count, const float* gscale,
                         const float* row_scale, hipStream_t st) {
  UG_REQUIRE(R > 0 && V > 0 && ld >= V && ld % 8 == 0, "ug_ce_bwd: need ld>=V and ld%%8==0");
  UG_REQUIRE(ug_aligned16(logits_inout) && labels && lse && (loss_and_count || row_scale), "ug_ce_bwd: bad pointers");
  hipLaunchKernelGGL(ce_bwd_kernel, dim3((unsigned)R), dim3(256), 0, st, (bf16_t*)logits_inout, ld, (int)V, labels,
                     ignore_index, lse, loss_and_count, gscale, row_scale);
  UG_CHECK_LAUNCH("ug_ce_bwd");
  return UG_OK;
}
