// MaskGIT parallel-decoding step (UniGen.t2i_generate, reference models/unigen.py:404-451 with
// models/sampling.py:24-46): classifier-free-guidance mix of the code-book logits, softmax, one categorical draw
// per image position, the confidence of the drawn token, and the re-masking of the least confident positions.
// Randomness is SUPPLIED as uniforms (drawn by the caller's generator), so a step is a deterministic function of
// its inputs: token = first index whose running sum of exp(logit - max) exceeds u * total (inverse CDF over the
// code book in index order); Gumbel noise = -log(-log(u2)).  torch.multinomial draws from the same distribution
// with a different use of the stream; the oracle restates this rule (oracle/qwen2_ref.py: InverseCdfSampler).
#include "common.h"
#include "unigen_hip.h"
#include "vmem_asm.h"

namespace {

constexpr int SMP_T = 256;

// one workgroup per (image, position); thread t owns code-book indices [t*C, (t+1)*C)
__global__ __launch_bounds__(SMP_T) void maskgit_sample_kernel(const bf16_t* __restrict__ logits, int64_t ld, int V, int N, int n,
                                                              int cfg, float scale, const float* __restrict__ u_sample,
                                                              const int64_t* __restrict__ cur_ids, int64_t mask_id,
                                                              int64_t* __restrict__ sampled, float* __restrict__ sel) {
  __shared__ float red[SMP_T / 64];
  __shared__ float part[SMP_T];
  __shared__ int hit;
  const int bi = blockIdx.x, t = threadIdx.x;
  const int64_t cur = cur_ids[bi];
  if (cur != mask_id) {                      // known token: kept, never re-masked (confidence = float max)
    if (t == 0) { sampled[bi] = cur; sel[bi] = 3.402823466e+38f; }
    return;
  }
  const int C = (V + SMP_T - 1) / SMP_T;
  const bf16_t* c = logits + (int64_t)bi * ld;
  const bf16_t* u = logits + ((int64_t)N * n + bi) * ld;
  const int lo = t * C, hi = min(V, lo + C);
  float mx = -INFINITY;
  for (int e = lo; e < hi; ++e) {
    const float cv = bf2f(c[e]);
    float v = cv;
    if (cfg) { const float uv = bf2f(u[e]); v = scale * (cv - uv) + uv; }
    mx = fmaxf(mx, v);
  }
  mx = block_max<SMP_T / 64>(mx, red);
  float local = 0.f;
  for (int e = lo; e < hi; ++e) {
    const float cv = bf2f(c[e]);
    float v = cv;
    if (cfg) { const float uv = bf2f(u[e]); v = scale * (cv - uv) + uv; }
    local += expf(v - mx);
  }
  part[t] = local;
  if (t == 0) hit = SMP_T - 1;
  __syncthreads();
  // exclusive prefix over the 256 chunk sums: every thread sums its predecessors (256 LDS reads, negligible)
  float excl = 0.f, total = 0.f;
  for (int j = 0; j < SMP_T; ++j) { const float pj = part[j]; if (j < t) excl += pj; total += pj; }
  const float target = u_sample[bi] * total;
  if (excl <= target && target < excl + local) atomicMin(&hit, t);      // first chunk whose range holds the target
  __syncthreads();
  if (t == hit) {
    float run = excl, ex = 0.f;
    int idx = hi - 1;
    for (int e = lo; e < hi; ++e) {
      const float cv = bf2f(c[e]);
      float v = cv;
      if (cfg) { const float uv = bf2f(u[e]); v = scale * (cv - uv) + uv; }
      ex = expf(v - mx);
      run += ex;
      if (run > target) { idx = e; break; }
    }
    if (idx == hi - 1 && !(run > target)) {       // rounding pushed the target past the end of the chunk / vocabulary
      const float cv = bf2f(c[idx]);
      float v = cv;
      if (cfg) { const float uv = bf2f(u[idx]); v = scale * (cv - uv) + uv; }
      ex = expf(v - mx);
    }
    sampled[bi] = idx;
    sel[bi] = ex / total;
  }
}

// one workgroup per image: confidence, rank of every position, threshold = (mask_len)-th smallest, re-mask below it
__global__ __launch_bounds__(1024) void maskgit_remask_kernel(const float* __restrict__ sel, const float* __restrict__ u_conf,
                                                              const int64_t* __restrict__ cur_ids, const int64_t* __restrict__ sampled,
                                                              int64_t mask_id, int64_t id_offset, int mask_len_sched,
                                                              float temperature, int n, int64_t* __restrict__ next_cur,
                                                              int64_t* __restrict__ next_ids, uint8_t* __restrict__ masking_out) {
  extern __shared__ float conf_s[];
  __shared__ int unknown_cnt;
  __shared__ float thr_s;
  const int b = blockIdx.x, i = threadIdx.x;
  if (i == 0) unknown_cnt = 0;
  __syncthreads();
  float conf = INFINITY;
  if (i < n) {
    const int64_t g = (int64_t)b * n + i;
    const float uu = fmaxf(u_conf[g], 1e-20f);
    const float gum = -logf(fmaxf(-logf(uu), 1e-20f));
    conf = logf(fmaxf(sel[g], 1e-20f)) + temperature * gum;
    conf_s[i] = conf;
    if (cur_ids[g] == mask_id) atomicAdd(&unknown_cnt, 1);
  }
  __syncthreads();
  const int k = max(1, min(unknown_cnt - 1, mask_len_sched));
  if (i < n) {
    int less = 0, eq = 0;
    for (int j = 0; j < n; ++j) { const float cj = conf_s[j]; less += cj < conf; eq += cj == conf; }
    if (less <= k && k < less + eq) thr_s = conf;        // sorted[k] (all writers hold the same value)
  }
  __syncthreads();
  if (i < n) {
    const int64_t g = (int64_t)b * n + i;
    const bool m = (k < n) && conf < thr_s;
    const int64_t s = sampled[g];
    next_cur[g] = m ? mask_id : s;
    next_ids[g] = m ? mask_id : s + id_offset;
    if (masking_out) masking_out[g] = m;
  }
}

// ------------------------------------------------------------------ autoregressive image-token sampling
// One decode step of UniGen.t2i_generate_ar after the backbone (reference models/unigen.py:503-519): code-book logits
// of the conditional / unconditional rows (raw fp32 accumulator of the lm-head GEMV, rounded to bf16 like the head's
// output), CFG mix, temperature, softmax, one categorical draw (inverse CDF on a supplied uniform) or argmax, then the
// next step's input: the embedding row of (token + id_offset) for both halves.  One workgroup per image; the
// accumulator rows are cleared for the next step.
constexpr int AR_MAXV = 8192;          // code-book slice staged in LDS (33 KiB incl. padding)
__device__ __forceinline__ int ar_pad(int e) { return e + (e >> 5); }      // chunk starts land on different banks

// 1 024 threads (round 5; was 256 with an O(threads) serial scan of the partial sums: 16.7 us per step): eight loads per row per
// thread, the inverse CDF through a wave scan (shuffles) + sixteen wave totals, the rest unchanged.
constexpr int ARS_T = 1024;
__global__ __launch_bounds__(ARS_T) void ar_sample_kernel(float* __restrict__ acc, int64_t lda, int bsz, int V, float scale,
                                                         float inv_temp, int greedy, const float* __restrict__ uniforms,
                                                         const int* __restrict__ pos_dev, int pos0, int nsteps,
                                                         const float* __restrict__ embed, int64_t lde, int H, int64_t id_offset,
                                                         int64_t* __restrict__ tok, int* __restrict__ out_tokens,
                                                         float* __restrict__ x) {
  __shared__ float mix[AR_MAXV + AR_MAXV / 32];
  __shared__ float red[ARS_T / 64];
  __shared__ float wtot[ARS_T / 64];
  __shared__ int hit;
  __shared__ int best_i[ARS_T / 64];
  __shared__ int chosen;
  const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  // the position word (written by the head launch in front of us): a hand-issued SCALAR load (vmem_asm.h: hipcc reads such a word with a
  // vector load and waits vmcnt(0) for it on the spot, a serialised round trip ahead of the logits' loads); used behind pass 1
  int pos_now;
  sld4(pos_now, pos_dev);
  float* c = acc + (int64_t)b * lda;
  float* u = acc + (int64_t)(bsz + b) * lda;
  // pass 1 (coalesced): mixed logits into LDS, accumulator rows cleared, running max / first argmax per thread
  float mx = -INFINITY;
  int arg = 0x7fffffff;
  {
    constexpr int PER = AR_MAXV / ARS_T;                     // every load of the two rows in flight at once
    float cr[PER], ur[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int e = min(t + j * ARS_T, V - 1);
      cr[j] = c[e]; ur[j] = u[e];
    }
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int e = t + j * ARS_T;
      if (e < V) {
        c[e] = 0.f; u[e] = 0.f;
        const float cv = bf2f(f2bf(cr[j])), uv = bf2f(f2bf(ur[j]));
        const float v = (uv + scale * (cv - uv)) * inv_temp;
        mix[ar_pad(e)] = v;
        if (v > mx) { mx = v; arg = e; }
      }
    }
  }
  wait_lgkm0();
  tie_s(pos_now);
  const int step = min(max(pos_now - pos0, 0), nsteps - 1);
  const float u01 = greedy ? 0.f : uniforms[(int64_t)step * bsz + b];      // (requested here: its round trip hides under the reductions)
  // block max + smallest index attaining it
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float om = __shfl_xor(mx, o, 64); const int oi = __shfl_xor(arg, o, 64);
    if (om > mx || (om == mx && oi < arg)) { mx = om; arg = oi; }
  }
  if (lane == 0) { red[wave] = mx; best_i[wave] = arg; }
  if (t == 0) hit = ARS_T - 1;
  __syncthreads();
  float bmx = red[0]; int bi = best_i[0];
#pragma unroll
  for (int w = 1; w < ARS_T / 64; ++w) if (red[w] > bmx || (red[w] == bmx && best_i[w] < bi)) { bmx = red[w]; bi = best_i[w]; }
  if (greedy) {
    if (t == 0) chosen = bi;
  } else {
    // inverse CDF in index order: thread t owns the contiguous chunk [t*C, (t+1)*C); exclusive prefix of the chunk sums by a
    // wave scan + the totals of the waves in front
    const int C = (V + ARS_T - 1) / ARS_T;
    const int lo = t * C, hi = min(V, lo + C);
    float local = 0.f;
    for (int e = lo; e < hi; ++e) local += expf(mix[ar_pad(e)] - bmx);
    float incl = local;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const float v = __shfl_up(incl, o, 64);
      if (lane >= o) incl += v;
    }
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    float base = 0.f, total = 0.f;
#pragma unroll
    for (int w = 0; w < ARS_T / 64; ++w) { const float wv = wtot[w]; if (w < wave) base += wv; total += wv; }
    const float excl = base + (incl - local);
    // The chunks' ranges must TILE [0, total): a thread's upper bound is its neighbour's lower bound, not its own excl + local (the two
    // differ by an ulp, so a target could fall into a gap, match nobody and force the token to the last chunk: ~3e-5 per token,
    // ADVICE r5).  Next lane's excl within the wave; the next wave's base (= base + this wave's total, the very sum that wave forms)
    // across waves; the grand total for the last thread.
    float upper = __shfl_down(excl, 1, 64);
    if (lane == 63) upper = (wave == ARS_T / 64 - 1) ? total : base + wtot[wave];
    const float target = u01 * total;
    if (excl <= target && target < upper) atomicMin(&hit, t);
    __syncthreads();
    if (t == hit) {
      float run = excl;
      int idx = max(hi - 1, lo);
      for (int e = lo; e < hi; ++e) { run += expf(mix[ar_pad(e)] - bmx); if (run > target) { idx = e; break; } }
      chosen = min(idx, V - 1);
    }
  }
  __syncthreads();
  const int token = chosen;
  if (t == 0) { tok[b] = token; out_tokens[(int64_t)b * nsteps + step] = token; }
  const float4* er = reinterpret_cast<const float4*>(embed + (token + id_offset) * lde);
  float4* x0 = reinterpret_cast<float4*>(x + (int64_t)b * H);
  float4* x1 = reinterpret_cast<float4*>(x + (int64_t)(bsz + b) * H);
  for (int i = t; i < (H >> 2); i += ARS_T) { const float4 v = er[i]; x0[i] = v; x1[i] = v; }
}

}  // namespace

extern "C" int ug_maskgit_step(const void* logits, int64_t ld, int64_t V, int64_t N, int64_t n, int cfg, float guidance_scale,
                               const float* u_sample, const float* u_conf, const int64_t* cur_ids, int64_t mask_id,
                               int64_t id_offset, int64_t mask_len_sched, float temperature, int64_t* sampled, float* sel_ws,
                               int64_t* next_cur, int64_t* next_ids, uint8_t* masking_out, hipStream_t st) {
  UG_REQUIRE(logits && u_sample && u_conf && cur_ids && sampled && sel_ws && next_cur && next_ids, "ug_maskgit_step: null argument");
  UG_REQUIRE(N > 0 && n > 0 && n <= 1024 && V > 0 && ld >= V, "ug_maskgit_step: bad sizes (N=%ld n=%ld V=%ld ld=%ld; n <= 1024)",
             (long)N, (long)n, (long)V, (long)ld);
  hipLaunchKernelGGL(maskgit_sample_kernel, dim3((unsigned)(N * n)), dim3(SMP_T), 0, st, (const bf16_t*)logits, ld, (int)V, (int)N,
                     (int)n, cfg, guidance_scale, u_sample, cur_ids, mask_id, sampled, sel_ws);
  UG_CHECK_LAUNCH("ug_maskgit_step(sample)");
  const int threads = (int)((n + 63) / 64 * 64);
  hipLaunchKernelGGL(maskgit_remask_kernel, dim3((unsigned)N), dim3(threads), n * sizeof(float), st, sel_ws, u_conf, cur_ids, sampled,
                     mask_id, id_offset, (int)mask_len_sched, temperature, (int)n, next_cur, next_ids, masking_out);
  UG_CHECK_LAUNCH("ug_maskgit_step(remask)");
  return UG_OK;
}

extern "C" int ug_ar_sample(float* acc, int64_t ldacc, int64_t bsz, int64_t V, float guidance_scale, float temperature, int greedy,
                            const float* uniforms, const int* pos_dev, int64_t pos0, int64_t nsteps, const float* embed,
                            int64_t ld_embed, int64_t H, int64_t id_offset, int64_t* tok, int* out_tokens, float* x, hipStream_t st) {
  UG_REQUIRE(acc && pos_dev && embed && tok && out_tokens && x && (greedy || uniforms), "ug_ar_sample: null argument");
  UG_REQUIRE(V <= AR_MAXV, "ug_ar_sample: code-book slice of %ld columns exceeds the %d this build stages in LDS", (long)V, AR_MAXV);
  UG_REQUIRE(bsz > 0 && V > 0 && ldacc >= V && nsteps > 0 && H > 0 && H % 4 == 0 && ld_embed % 4 == 0 && temperature > 0.f &&
                 ug_aligned16(embed) && ug_aligned16(x),
             "ug_ar_sample: bad sizes (bsz=%ld V=%ld H=%ld temperature=%g)", (long)bsz, (long)V, (long)H, (double)temperature);
  hipLaunchKernelGGL(ar_sample_kernel, dim3((unsigned)bsz), dim3(ARS_T), 0, st, acc, ldacc, (int)bsz, (int)V, guidance_scale,
                     1.f / temperature, greedy, uniforms, pos_dev, (int)pos0, (int)nsteps, embed, ld_embed, (int)H, id_offset, tok,
                     out_tokens, x);
  UG_CHECK_LAUNCH("ug_ar_sample");
  return UG_OK;
}
