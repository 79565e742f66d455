// Autoregressive decode support for UniGen.t2i_generate_ar (reference models/unigen.py:457-521, which
// drives transformers' DynamicCache + SDPA one token at a time).  Static KV cache sized for the whole
// generation, position / length read from DEVICE memory so one captured hipGraph replays all 256 steps.
//   cache layout: K,V [rows][HKV][Tmax][128] bf16 (keys of one (row, kv-head) contiguous)
#include "common.h"
#include "unigen_hip.h"
#include "vmem_asm.h"
#include <stdlib.h>

namespace {

constexpr int DHD = 128;
// workgroup barrier for an LDS hand-off: __syncthreads() also drains vmcnt (its fence covers global memory), i.e. waits for the
// weight tiles in flight and the clears' write acknowledgements, which no wave needs at this point
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
typedef const __attribute__((address_space(1))) void* gptr_t;      // LDS-DMA operands (global_load_lds)
typedef __attribute__((address_space(3))) void* lptr_t;

// qkv rows (r*L + t) -> cache[r][hk][pos0 + t][:]   (k already rotated).  One thread per 16-byte chunk.
__global__ __launch_bounds__(256) void kv_store_kernel(const bf16_t* __restrict__ qkv, int64_t ldq, int k_col, int v_col,
                                                       bf16_t* __restrict__ ck, bf16_t* __restrict__ cv, int R, int L,
                                                       int HKV, int Tmax, const int* __restrict__ pos_dev, int pos_host) {
  const int pos0 = pos_dev ? *pos_dev : pos_host;
  const int64_t total = (int64_t)R * L * HKV * (DHD / 8);
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % (DHD / 8));
    const int hk = (int)((idx / (DHD / 8)) % HKV);
    const int64_t tok = idx / ((DHD / 8) * HKV);
    const int r = (int)(tok / L), t = (int)(tok % L);
    if (pos0 + t >= Tmax) continue;
    const bf16_t* src = qkv + tok * ldq + hk * DHD + c * 8;
    const int64_t dst = (((int64_t)r * HKV + hk) * Tmax + pos0 + t) * DHD + c * 8;
    *reinterpret_cast<bf16x8_t*>(ck + dst) = *reinterpret_cast<const bf16x8_t*>(src + k_col);
    *reinterpret_cast<bf16x8_t*>(cv + dst) = *reinterpret_cast<const bf16x8_t*>(src + v_col);
  }
}

// RoPE for single-token rows at position *pos_dev (same arithmetic as rope_kernel in elementwise.hip)
__global__ __launch_bounds__(256) void rope_at_kernel(bf16_t* __restrict__ qkv, const float* __restrict__ cs,
                                                      const float* __restrict__ sn, int rows, int ldq, int nheads, int hd,
                                                      const int* __restrict__ pos_dev, int max_pos) {
#pragma clang fp contract(off)
  const int half = hd >> 1, per_head = half >> 2;
  const int pos = min(*pos_dev, max_pos - 1);
  const int total = rows * nheads * per_head;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int p4 = idx % per_head;
    const int h = (idx / per_head) % nheads;
    const int t = idx / (per_head * nheads);
    bf16_t* base = qkv + (int64_t)t * ldq + h * hd + p4 * 4;
    const uint2 lo = *reinterpret_cast<const uint2*>(base);
    const uint2 hi = *reinterpret_cast<const uint2*>(base + half);
    const float4 c = *reinterpret_cast<const float4*>(cs + (int64_t)pos * half + p4 * 4);
    const float4 s = *reinterpret_cast<const float4*>(sn + (int64_t)pos * half + p4 * 4);
    const float x1[4] = {bf2f(lo.x & 0xffff), bf2f(lo.x >> 16), bf2f(lo.y & 0xffff), bf2f(lo.y >> 16)};
    const float x2[4] = {bf2f(hi.x & 0xffff), bf2f(hi.x >> 16), bf2f(hi.y & 0xffff), bf2f(hi.y >> 16)};
    const float cc[4] = {c.x, c.y, c.z, c.w}, ss[4] = {s.x, s.y, s.z, s.w};
    float o1[4], o2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float a1 = x1[k] * cc[k], a2 = x2[k] * cc[k];
      const float b1 = x2[k] * ss[k], b2 = x1[k] * ss[k];
      o1[k] = a1 - b1; o2[k] = a2 + b2;
    }
    uint2 olo, ohi;
    olo.x = pack_bf2(o1[0], o1[1]); olo.y = pack_bf2(o1[2], o1[3]);
    ohi.x = pack_bf2(o2[0], o2[1]); ohi.y = pack_bf2(o2[2], o2[3]);
    *reinterpret_cast<uint2*>(base) = olo;
    *reinterpret_cast<uint2*>(base + half) = ohi;
  }
}

// One workgroup (4 waves) per (row, query head).  Wave w takes key chunks w, w+4, ... of 64 keys: lane j scores
// key (t0 + j) against q (fp32 in LDS) and the wave keeps its own online-softmax state; for P.V the lanes regroup
// as (key phase = lane>>4, 8-dim chunk = lane&15) so every V read is a 16-byte load with 16 of them in flight.
// The four partial (m, l, O) states are merged through LDS (flash-decoding style split over keys).
// key_valid: optional [rows][Tmax] bytes (0 = padding key, never attended), like HF's 2-D attention_mask.
constexpr int AD_WAVES = 4;
__global__ __launch_bounds__(64 * AD_WAVES) void attn_decode_kernel(const bf16_t* __restrict__ q, int64_t ldq, const bf16_t* __restrict__ ck,
                                                          const bf16_t* __restrict__ cv, const uint8_t* __restrict__ key_valid,
                                                          bf16_t* __restrict__ o, int64_t ldo, int H, int HKV, int Tmax,
                                                          const int* __restrict__ len_dev, float scale) {
  __shared__ float qs[DHD];
  __shared__ float om[AD_WAVES][DHD];
  __shared__ float ml[AD_WAVES][2];
  const int r = blockIdx.y, h = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int hk = h / (H / HKV);
  const int len = min(*len_dev, Tmax);
  const bf16_t* qp = q + (int64_t)r * ldq + h * DHD;
  if (threadIdx.x < DHD) qs[threadIdx.x] = bf2f(qp[threadIdx.x]);
  __syncthreads();
  const bf16_t* kb = ck + ((int64_t)r * HKV + hk) * Tmax * DHD;
  const bf16_t* vb = cv + ((int64_t)r * HKV + hk) * Tmax * DHD;
  const int kq = lane >> 4, dc = lane & 15;
  float m = -INFINITY, l = 0.f;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  for (int t0 = wave * 64; t0 < len; t0 += 64 * AD_WAVES) {
    const int t = t0 + lane;
    float s = -INFINITY;
    // K row of this lane's key and the V pieces of its (key phase, dim chunk) all go out before any is consumed
    bf16x8_t kf[DHD / 8], vf[16];
    {
      const bf16_t* kr = kb + (int64_t)min(t, len - 1) * DHD;
#pragma unroll
      for (int c = 0; c < DHD / 8; ++c) kf[c] = *reinterpret_cast<const bf16x8_t*>(kr + c * 8);
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) {
        const int tt = min(t0 + jj * 4 + kq, len - 1);
        vf[jj] = *reinterpret_cast<const bf16x8_t*>(vb + (int64_t)tt * DHD + dc * 8);
      }
    }
    if (t < len && (!key_valid || key_valid[(int64_t)r * Tmax + t])) {
      float d = 0.f;
#pragma unroll
      for (int c = 0; c < DHD / 8; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) d += bf2f((bf16_t)kf[c][e]) * qs[c * 8 + e];
      s = d * scale;
    }
    const float mc = wave_max(s);
    const float mn = fmaxf(m, mc);
    const float mu = (mn == -INFINITY) ? 0.f : mn;
    const float alpha = __expf(m - mu);
    const float p = __expf(s - mu);
    l = l * alpha + wave_sum(p);
    m = mn;
    const float pb = bf2f(f2bf(p));          // P is rounded to bf16 before P.V like the bf16 SDPA paths
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] *= alpha;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
      const float pj = __shfl(pb, jj * 4 + kq, 64);   // 0 for keys past len / masked keys
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += pj * bf2f((bf16_t)vf[jj][e]);
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    acc[e] += __shfl_xor(acc[e], 16, 64);
    acc[e] += __shfl_xor(acc[e], 32, 64);
  }
  if (lane < 16) {
#pragma unroll
    for (int e = 0; e < 8; ++e) om[wave][dc * 8 + e] = acc[e];
  }
  if (lane == 0) { ml[wave][0] = m; ml[wave][1] = l; }
  __syncthreads();
  if (threadIdx.x < DHD) {
    float M = ml[0][0];
#pragma unroll
    for (int w = 1; w < AD_WAVES; ++w) M = fmaxf(M, ml[w][0]);
    float L = 0.f, O = 0.f;
#pragma unroll
    for (int w = 0; w < AD_WAVES; ++w) {
      const float wgt = (ml[w][0] == -INFINITY) ? 0.f : __expf(ml[w][0] - M);
      L += wgt * ml[w][1];
      O += wgt * om[w][threadIdx.x];
    }
    const float inv = L > 0.f ? 1.f / L : 0.f;
    o[(int64_t)r * ldo + h * DHD + threadIdx.x] = f2bf(O * inv);
  }
}

typedef __bf16 bf16pair_t __attribute__((ext_vector_type(2)));
constexpr int ADF_WAVES = 8;            // one 64-key chunk per wave up to 512 keys: no serialized second chunk
// Cache attention of one decode step fed by the RAW qkv accumulator: every (row, query head) workgroup rebuilds
// q and its kv head's new k / v row itself (no dependency between workgroups; the first query head of each kv head
// also appends the row to the cache), attends to cache keys [0, pos) exactly like attn_decode_kernel, and merges
// the new token as one more (m, l, O) partial.
// (Argument order: the first 14 dwords are what the kernel's first loads need -- with -mllvm -amdgpu-kernarg-preload-count=16 hipcc preloads
// that many, they arrive in SGPRs with the wave; the rest comes by one batch of scalar loads.)
__global__ __launch_bounds__(64 * ADF_WAVES) void attn_decode_fused_kernel(
    const float* __restrict__ acc_qkv, const float* __restrict__ ss, const int* __restrict__ pos_dev, bf16_t* __restrict__ ck,
    bf16_t* __restrict__ cv, int R, int HKV, int H, int lda, int Tmax, int max_pos,
    const bf16_t* __restrict__ bias, const float* __restrict__ cs, const float* __restrict__ sn, const uint8_t* __restrict__ key_valid,
    bf16_t* __restrict__ o, int64_t ldo, float eps, int norm_cols, float scale) {
  __shared__ float qs[DHD], kn[DHD], vn[DHD];
  __shared__ __attribute__((aligned(16))) bf16_t qb[DHD];       // q again, as bf16 pairs (q IS bf16-rounded): the B operand of v_dot2c_f32_bf16
  __shared__ float om[ADF_WAVES][DHD];
  __shared__ float ml[ADF_WAVES][2];
  // K rows of a wave's 64-key chunk: 16 KB, CONTIGUOUS in the cache ([row][kv head][t][128]).  Round 3 let lane j load "its" key
  // row piece by piece (16 loads of 16 B at a 256-byte lane stride): every instruction touched 64 cache lines and the 256 KB a
  // workgroup walks do not stay in the CU's 32 KB L1, so each line came up from L2 up to eight times -- 5.3 of the kernel's
  // 10.7 us (ablation, profiles/r04_ar_decode.md).  Now the chunk goes HBM/L2 -> LDS by LDS-DMA, 1 KB of contiguous source per
  // instruction, and lane j reads its row back from LDS.  LDS-DMA writes lane-linear, so the bank swizzle is applied on the
  // SOURCE side: LDS slot s of key k holds 16-byte chunk s ^ (k & 15); the 16 lanes of a ds_read_b128 service group hold 16
  // different k & 15, so the reads are conflict-free.
  __shared__ __attribute__((aligned(1024))) char ktile[ADF_WAVES][64 * DHD * 2];
  // Workgroup -> (row, query head): the H / HKV query heads that share one (row, kv head)'s K / V are placed on ONE XCD (workgroup
  // b runs on XCD b % 8 and every XCD has its own L2): dealt out head-major they landed on six different XCDs and each XCD
  // fetched the same K / V from HBM -- 36 MB per layer instead of 6, the part of this kernel that grew with the context
  // (8.2 us at 171 keys, 11.2 at 363).
  // Grid (8, H / HKV, ceil(R HKV / 8)): x = XCD, y = query head inside the kv group, z = block of eight groups -- the linear
  // workgroup id is what it was (8 (z per + y) + x), and no division is left ahead of the first load (round 5).
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int hq = blockIdx.y;
  const int grp = blockIdx.x + 8 * blockIdx.z;            // (row, kv head) group
  // Arguments in two batches (vmem_asm.h, argument hoisting): the early exit, the power-of-two test and the position load used to be three
  // DEPENDENT round trips to the kernarg segment ahead of the first vector load.  Batch 1 = the kernel's first 14 argument dwords --
  // everything the accumulator loads (the ~2 us chain) need -- which arrive in SGPRs with the wave when the
  // code object is built with kernarg preload; batch 2 is requested behind the accumulator loads.
  asm volatile("" ::"s"(acc_qkv), "s"(ss), "s"(pos_dev), "s"(ck), "s"(cv), "s"(R), "s"(HKV), "s"(H), "s"(lda));
  if (grp >= R * HKV) return;
  int r, hk, per;
  if ((HKV & (HKV - 1)) == 0) { const int sh = 31 - __builtin_clz(HKV); r = grp >> sh; hk = grp & (HKV - 1); per = H >> sh; }
  else { r = grp / HKV; hk = grp - r * HKV; per = H / HKV; }          // (= gridDim.y, but that is an implicit ARGUMENT: one more scalar load)
  const int h = hk * per + hq;
  const int kq = lane >> 4, dc = lane & 15;
  const bf16_t *kb, *vb;                                   // this (row, kv head)'s cache rows: set behind the second argument batch
  // the first key chunk's K rows / V pieces do not depend on the new token: request them before the prologue's own
  // round trip (accumulator, bias, RoPE table) so the two latencies overlap.  (Requesting them before the position word
  // as well -- clamped to the cache instead of the visible length, masked afterwards -- measured SLOWER, 5 730 vs 5 885
  // tokens/s: every wave then loads a chunk at every step, visible or not.)
  // Every load of this kernel is hand-issued (vmem_asm.h): with the builtins the prologue's LDS writes were preceded by a compiler-inserted
  // s_waitcnt vmcnt(0), i.e. the new token's q / k / v were built only after the wave's whole K / V chunk had landed.
  bf16x8_t vf[16];
  const uint32_t kt_lds = __builtin_amdgcn_readfirstlane(lds_addr_of(ktile[wave]));
  auto load_chunk = [&](int t0, int last) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int k = i * 4 + kq;                                    // key of the chunk this lane's 16 bytes belong to
      const bf16_t* src = kb + (int64_t)min(t0 + k, last) * DHD + ((dc ^ (k & 15)) << 3);
      dma16(src, kt_lds + i * 1024);
    }
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
      const int tt = min(t0 + jj * 4 + kq, last);
      ld16(vf[jj], vb + (int64_t)tt * DHD + dc * 8);
    }
  };
  int pos0;
  sld4(pos0, pos_dev);                                     // (hand-issued scalar load: vmem_asm.h)
  // The new token's q / k / v pair of this lane (waves 0-2): its loads -- the raw accumulator the projection's atomics left at
  // the device coherence point, bias, RoPE table: one ~2 us round trip -- go out FIRST, the chunk's K / V requests behind them.
  const float* arow = acc_qkv + (int64_t)r * lda;
  const int col0 = wave == 0 ? h * DHD : wave == 1 ? (H + hk) * DHD : (H + HKV + hk) * DHD;
  float a1 = 0.f, a2 = 0.f, rc = 1.f, rsn = 0.f, ssr = 1.f;
  uint32_t bb1 = 0, bb2 = 0;
  if (wave < 3) {
    ld4(ssr, ss + r);
    ld4(a1, arow + col0 + lane); ld4(a2, arow + col0 + lane + DHD / 2);
  }
  asm volatile("" ::"s"(Tmax), "s"(max_pos), "s"(bias), "s"(cs), "s"(sn), "s"(key_valid), "s"(o), "s"(ldo), "s"(__float_as_int(eps)),
               "s"(norm_cols), "s"(__float_as_int(scale)));  // batch 2
  wait_lgkm0();                                            // (the position word; the batch above is behind the same counter)
  tie_s(pos0);
  kb = ck + ((int64_t)r * HKV + hk) * Tmax * DHD;
  vb = cv + ((int64_t)r * HKV + hk) * Tmax * DHD;
  const int len = min(pos0, Tmax);                         // cache keys visible to the new token
  const int rpos = min(pos0, max_pos - 1);
  if (wave < 3) {
    if (bias) { ld2u(bb1, bias + col0 + lane); ld2u(bb2, bias + col0 + lane + DHD / 2); }
    if (wave < 2) { ld4(rc, cs + (int64_t)rpos * (DHD / 2) + lane); ld4(rsn, sn + (int64_t)rpos * (DHD / 2) + lane); }
  }
  const bool has_chunk = wave * 64 < len;
  if (has_chunk) load_chunk(wave * 64, len - 1);
  if (wave < 3) {
    if (has_chunk) wait_vm<32>(); else wait_vm<0>();       // the prologue's loads are older than the chunk's 16 + 16
    tie(ssr); tie(a1); tie(a2); tie(bb1); tie(bb2); tie(rc); tie(rsn);
    const float b1 = bf2f((bf16_t)bb1), b2 = bf2f((bf16_t)bb2);
    float x1, x2;
    {
#pragma clang fp contract(off)
      // consumer-side finishing of the qkv projection, same arithmetic as finish_qkv_tile / rope_at_kernel: bf16(rstd * acc + bias), rotate-half with
      // separately rounded products
      const float rs = rsqrtf(ssr / (float)norm_cols + eps);
      float v1 = rs * a1, v2 = rs * a2;
      if (bias) { v1 += b1; v2 += b2; }
      x1 = bf2f(f2bf(v1)); x2 = bf2f(f2bf(v2));
      if (wave < 2) {
        const float p1 = x1 * rc, p2 = x2 * rc;
        const float q1 = x2 * rsn, q2 = x1 * rsn;
        x1 = bf2f(f2bf(p1 - q1)); x2 = bf2f(f2bf(p2 + q2));
      }
    }
    float* dst = wave == 0 ? qs : wave == 1 ? kn : vn;
    dst[lane] = x1; dst[lane + DHD / 2] = x2;
    if (wave == 0) { qb[lane] = f2bf(x1); qb[lane + DHD / 2] = f2bf(x2); }
    if (wave > 0 && hq == 0 && pos0 < Tmax) {
      bf16_t* row = (wave == 1 ? ck : cv) + (((int64_t)r * HKV + hk) * Tmax + pos0) * DHD;
      row[lane] = f2bf(x1); row[lane + DHD / 2] = f2bf(x2);
    }
  }
  lds_barrier();                                           // (LDS hand-off only: __syncthreads() would also drain the chunk's loads)
  float m = -INFINITY, l = 0.f;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll 1
  for (int t0 = wave * 64; t0 < len; t0 += 64 * ADF_WAVES) {
    const int t = t0 + lane;
    float s = -INFINITY;
    if (t0 != wave * 64) {                               // later chunks (contexts beyond 512 keys): the wave's own reads of the
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); //  previous chunk have returned before the DMA overwrites the tile
      load_chunk(t0, len - 1);
    }
    wait_vm<0>();                                        // the chunk's K rows have landed in LDS (this wave's own DMA), its V pieces in vf
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) tie(vf[jj]);
    if (t < len && (!key_valid || key_valid[(int64_t)r * Tmax + t])) {
      // 64 v_dot2c_f32_bf16 (two bf16 products + fp32 add each) instead of 128 conversions + 128 fmas: K pairs straight from the
      // tile, q pairs from its bf16 image (broadcast reads: hoisting all 64 pairs would not fit 8 waves' registers)
      float d = 0.f;
      const char* krow = ktile[wave] + lane * (DHD * 2);
#pragma unroll
      for (int c = 0; c < DHD / 8; ++c) {
        const uint4 kf = *reinterpret_cast<const uint4*>(krow + ((c ^ (lane & 15)) << 4));
        const uint4 qq = *reinterpret_cast<const uint4*>(&qb[c * 8]);
        d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16pair_t, kf.x), __builtin_bit_cast(bf16pair_t, qq.x), d, false);
        d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16pair_t, kf.y), __builtin_bit_cast(bf16pair_t, qq.y), d, false);
        d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16pair_t, kf.z), __builtin_bit_cast(bf16pair_t, qq.z), d, false);
        d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16pair_t, kf.w), __builtin_bit_cast(bf16pair_t, qq.w), d, false);
      }
      s = d * scale;
    }
    const float mc = wave_max(s);
    const float mn = fmaxf(m, mc);
    const float mu = (mn == -INFINITY) ? 0.f : mn;
    const float alpha = __expf(m - mu);
    const float p = __expf(s - mu);
    l = l * alpha + wave_sum(p);
    m = mn;
    const float pb = bf2f(f2bf(p));
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] *= alpha;
    // P.V two keys at a time: the two keys' values of a dim pair are interleaved by v_perm_b32 into (v[j][d], v[j+1][d]) and meet
    // the probability pair (p_j, p_j+1; both bf16-exact) in one v_dot2c_f32_bf16 -- 8 perm + 8 dot2 per key pair instead of 16
    // conversions + 16 fmas
#pragma unroll
    for (int jj = 0; jj < 16; jj += 2) {
      const uint32_t pp = pack_bf2(__shfl(pb, jj * 4 + kq, 64), __shfl(pb, (jj + 1) * 4 + kq, 64));
      const uint4 va = __builtin_bit_cast(uint4, vf[jj]), vb = __builtin_bit_cast(uint4, vf[jj + 1]);
      const uint32_t wa[4] = {va.x, va.y, va.z, va.w}, wb[4] = {vb.x, vb.y, vb.z, vb.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint32_t lo = __builtin_amdgcn_perm(wb[i], wa[i], 0x05040100u);      // (a.lo16, b.lo16): dim 2 i of keys j, j + 1
        const uint32_t hi = __builtin_amdgcn_perm(wb[i], wa[i], 0x07060302u);      // (a.hi16, b.hi16): dim 2 i + 1
        acc[2 * i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16pair_t, lo), __builtin_bit_cast(bf16pair_t, pp), acc[2 * i], false);
        acc[2 * i + 1] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16pair_t, hi), __builtin_bit_cast(bf16pair_t, pp), acc[2 * i + 1], false);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    acc[e] += __shfl_xor(acc[e], 16, 64);
    acc[e] += __shfl_xor(acc[e], 32, 64);
  }
  if (lane < 16) {
#pragma unroll
    for (int e = 0; e < 8; ++e) om[wave][dc * 8 + e] = acc[e];
  }
  if (lane == 0) { ml[wave][0] = m; ml[wave][1] = l; }
  // the new token's own score (every wave computes it; no extra barrier)
  const float s_new = wave_sum(qs[lane] * kn[lane] + qs[lane + DHD / 2] * kn[lane + DHD / 2]) * scale;
  __syncthreads();
  if (threadIdx.x < DHD) {
    float M = s_new;
#pragma unroll
    for (int w = 0; w < ADF_WAVES; ++w) M = fmaxf(M, ml[w][0]);
    const float wn = __expf(s_new - M);
    float L = wn, O = wn * vn[threadIdx.x];
#pragma unroll
    for (int w = 0; w < ADF_WAVES; ++w) {
      const float wgt = (ml[w][0] == -INFINITY) ? 0.f : __expf(ml[w][0] - M);
      L += wgt * ml[w][1];
      O += wgt * om[w][threadIdx.x];
    }
    o[(int64_t)r * ldo + h * DHD + threadIdx.x] = f2bf(O / L);
  }
}

// Finish a split-K fp32 accumulation of a skinny GEMM:  mode 0: out_bf16 = bf16(acc + bias)
//                                                      mode 1: resid_f32 += bf16round(acc)      (in place)
__global__ __launch_bounds__(256) void skinny_finish_kernel(const float* __restrict__ acc, const bf16_t* __restrict__ bias,
                                                            bf16_t* __restrict__ out_bf16, float* __restrict__ resid,
                                                            int64_t total, int N, int mode) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    float v = acc[i];
    if (mode == 0) {
      if (bias) v += bf2f(bias[i % N]);
      out_bf16[i] = f2bf(v);
    } else {
      resid[i] += bf2f(f2bf(v));
    }
  }
}

// ------------------------------------------------------------------ weight-streaming GEMV for decode
// acc[r][n] += sum_k x[r][k] W[n][k]   for a handful of rows r (<= 16*RB) and a weight matrix read ONCE.
// HBM-bound and latency-sensitive (a whole projection is a few MB): the launch is cut into as many waves as
// it takes for (nearly) the whole matrix to be in flight at once.  One wave owns 16 weight rows x 32*U k,
// issues its U 16-byte non-temporal loads back to back, then feeds mfma_f32_16x16x32_bf16 with A = activation
// fragment (a few KiB, L2-resident) and B = weight fragment, so a lane ends up holding 4 rows x 1 column and
// every atomic instruction of a wave covers 4 rows x 64 contiguous bytes of a ROW-MAJOR accumulator.  The 4 waves of a workgroup
// take consecutive k-slices of the same 16 columns, reduce through LDS, then one fp32 atomic per output.
// acc is addressed acc[r*sr + n*sn]; the decode path uses row-major accumulators (sn = 1).
template <int RB, int U>
__global__ __launch_bounds__(256) void gemv_kernel(const bf16_t* __restrict__ x, int64_t ldx, int R,
                                                   const bf16_t* __restrict__ W, int64_t ldw, float* __restrict__ acc,
                                                   int64_t sr, int64_t sn, int N, int K) {
  __shared__ float red[3][64][4 * RB];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
  const int n0 = blockIdx.x * 16;
  const int nrow = min(n0 + (lane & 15), N - 1);
  const int kbase = (blockIdx.y * 4 + wave) * (32 * U);
  const bf16_t* wp = W + (int64_t)nrow * ldw + g * 8 + kbase;
  f32x4_t d[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) d[rb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  bf16x8_t wf[U];
#pragma unroll
  for (int u = 0; u < U; ++u)
    if (kbase + u * 32 < K) wf[u] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8_t*>(wp + u * 32));
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const bf16_t* xp = x + (int64_t)min(rb * 16 + (lane & 15), R - 1) * ldx + g * 8 + kbase;
    bf16x8_t xf[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (kbase + u * 32 < K) xf[u] = *reinterpret_cast<const bf16x8_t*>(xp + u * 32);
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (kbase + u * 32 < K) d[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[u], wf[u], d[rb], 0, 0, 0);
  }
  if (wave > 0) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int j = 0; j < 4; ++j) red[wave - 1][lane][rb * 4 + j] = d[rb][j];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const int n = n0 + (lane & 15);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = rb * 16 + g * 4 + j;
        const float v = d[rb][j] + red[0][lane][rb * 4 + j] + red[1][lane][rb * 4 + j] + red[2][lane][rb * 4 + j];
        if (r < R && n < N) atomicAdd(acc + (int64_t)r * sr + (int64_t)n * sn, v);
      }
    }
  }
}

// Main form (K >= 256: every projection of the backbone and the lm-head slices).  Per-instruction coalescing
// decides the achieved HBM rate, and the MFMA operand layout (lane = weight row) only lets a direct load touch
// 64 bytes per row (measured 2.8 TB/s on gate_up).  So weight tiles go HBM -> LDS with 16-byte LDS-DMA, each
// instruction covering two rows x 512 contiguous bytes, and the fragments come back out of LDS.  LDS-DMA writes
// lane-linear, so the bank swizzle (16-byte chunk index ^ row) is applied to the SOURCE address.

// ------------------------------------------------------------------ fused finishers of the decode step
// Every finisher consumes a row-major fp32 accumulator filled by the GEMV (acc[r*lda + n]) and leaves it ZEROED
// for the next step, so the captured graph carries no memset nodes.

// q/k/v projection: + bias -> bf16, RoPE at *pos_dev on the q and k heads, q -> q_out, k/v -> cache[pos].
// One wave per (head, row); lane i owns dims i and i+64 (the rotary pair).  Same arithmetic as rope_at_kernel.
__device__ __forceinline__ void finish_qkv_tile(float* __restrict__ acc, int64_t lda, const bf16_t* __restrict__ bias,
                                                const float* __restrict__ cs, const float* __restrict__ sn_tab, int pos0,
                                                bf16_t* __restrict__ q_out, int64_t ldq, bf16_t* __restrict__ ck,
                                                bf16_t* __restrict__ cv, int Hq, int Hk, int Tmax, int max_pos, int hh, int r,
                                                int i) {
#pragma clang fp contract(off)
  const int c1 = hh * DHD + i, c2 = c1 + DHD / 2;
  float* a = acc + (int64_t)r * lda;
  float v1 = a[c1], v2 = a[c2];
  a[c1] = 0.f; a[c2] = 0.f;
  if (bias) { v1 += bf2f(bias[c1]); v2 += bf2f(bias[c2]); }
  float x1 = bf2f(f2bf(v1)), x2 = bf2f(f2bf(v2));
  if (hh < Hq + Hk) {
    const int pos = min(pos0, max_pos - 1);
    const float c = cs[(int64_t)pos * (DHD / 2) + i], s = sn_tab[(int64_t)pos * (DHD / 2) + i];
    const float p1 = x1 * c, p2 = x2 * c;
    const float q1 = x2 * s, q2 = x1 * s;
    x1 = bf2f(f2bf(p1 - q1)); x2 = bf2f(f2bf(p2 + q2));
  }
  if (hh < Hq) {
    bf16_t* qp = q_out + (int64_t)r * ldq + hh * DHD;
    qp[i] = f2bf(x1); qp[i + DHD / 2] = f2bf(x2);
  } else if (pos0 < Tmax) {
    const bool is_k = hh < Hq + Hk;
    const int hk = is_k ? hh - Hq : hh - Hq - Hk;
    bf16_t* dst = (is_k ? ck : cv) + (((int64_t)r * Hk + hk) * Tmax + pos0) * DHD;
    dst[i] = f2bf(x1); dst[i + DHD / 2] = f2bf(x2);
  }
}

// o / down projection:  x[r] += bf16round(acc[r]);  xn[r] = bf16(rmsnorm(x[r]) * w)   (next sub-block's input).
// One wave per row, the same per-lane accumulation order as rmsnorm_fwd_kernel so both paths agree bit for bit.
template <int MAXV>
__device__ __forceinline__ void finish_resid_norm_row(float* __restrict__ acc, int64_t lda, float* __restrict__ x,
                                                      const float* __restrict__ w, bf16_t* __restrict__ xn, int cols, float eps,
                                                      int row, int lane) {
  float4* xr = reinterpret_cast<float4*>(x + (int64_t)row * cols);
  float4* ar = reinterpret_cast<float4*>(acc + (int64_t)row * lda);
  const float4* wr = reinterpret_cast<const float4*>(w);
  const int nv = cols >> 2;
  float4 v[MAXV], gw[MAXV], dl[MAXV];
#pragma unroll
  for (int c = 0; c < MAXV; ++c) {              // every load of the row in flight at once
    const int i = lane + c * 64;
    if (i < nv) { v[c] = xr[i]; gw[c] = wr[i]; dl[c] = ar[i]; }
  }
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < MAXV; ++c) {
    const int i = lane + c * 64;
    if (i < nv) {
      ar[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      v[c].x += bf2f(f2bf(dl[c].x)); v[c].y += bf2f(f2bf(dl[c].y));
      v[c].z += bf2f(f2bf(dl[c].z)); v[c].w += bf2f(f2bf(dl[c].w));
      xr[i] = v[c];
      ss += v[c].x * v[c].x + v[c].y * v[c].y + v[c].z * v[c].z + v[c].w * v[c].w;
    }
  }
  ss = wave_sum(ss);
  const float rs = rsqrtf(ss / (float)cols + eps);
#pragma unroll
  for (int c = 0; c < MAXV; ++c) {
    const int i = lane + c * 64;
    if (i < nv) {
      uint2 o;
      o.x = pack_bf2(gw[c].x * (v[c].x * rs), gw[c].y * (v[c].y * rs));
      o.y = pack_bf2(gw[c].z * (v[c].z * rs), gw[c].w * (v[c].w * rs));
      reinterpret_cast<uint2*>(xn + (int64_t)row * cols)[i] = o;
    }
  }
}

template <int MAXV>
__global__ __launch_bounds__(64) void finish_resid_norm_kernel(float* __restrict__ acc, int64_t lda, float* __restrict__ x,
                                                               const float* __restrict__ w, bf16_t* __restrict__ xn, int cols,
                                                               float eps, int* __restrict__ pos_inc, int* __restrict__ len_inc) {
  finish_resid_norm_row<MAXV>(acc, lda, x, w, xn, cols, eps, blockIdx.x, threadIdx.x);
  // the step's last launch that comes after every reader of the write position: advance it here (was two one-element torch
  // kernels per decode step)
  if (pos_inc && blockIdx.x == 0 && threadIdx.x == 0) { ++*pos_inc; ++*len_inc; }
}

// gate/up projection:  act = bf16( bf16(silu(bf16 gate)) * bf16 up )  (swiglu_fwd_kernel's arithmetic).
__device__ __forceinline__ void finish_swiglu_quad(float* __restrict__ acc, int64_t lda, bf16_t* __restrict__ act, int I, int r,
                                                   int c) {
  float4* ag = reinterpret_cast<float4*>(acc + (int64_t)r * lda + c);
  float4* au = reinterpret_cast<float4*>(acc + (int64_t)r * lda + I + c);
  const float4 gv = *ag, uv = *au;
  *ag = make_float4(0.f, 0.f, 0.f, 0.f);
  *au = make_float4(0.f, 0.f, 0.f, 0.f);
  const float gs[4] = {gv.x, gv.y, gv.z, gv.w}, us[4] = {uv.x, uv.y, uv.z, uv.w};
  float o[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float g = bf2f(f2bf(gs[e])), u = bf2f(f2bf(us[e]));
    o[e] = bf2f(f2bf(g / (1.f + __expf(-g)))) * u;
  }
  uint2 ov; ov.x = pack_bf2(o[0], o[1]); ov.y = pack_bf2(o[2], o[3]);
  *reinterpret_cast<uint2*>(act + (int64_t)r * I + c) = ov;
}

// One wave per workgroup owns ONE 256-wide k-slab and KW consecutive 16-row weight groups.  The
// activation fragments of the slab are built once and stay in registers; the KW weight tiles walk through a
// two-slot LDS ring (tile t+2's DMA is issued as soon as tile t's fragments are in registers).  No workgroup
// barrier, no cross-wave reduction; ~10 independent waves per CU keep 16 KiB each in flight.
//
// Decode fusion.  A decode step is bound by the ~4 us floor of every launch, not by bytes, and on this multi-XCD
// part any in-kernel cross-wave hand-off costs several trips to the device coherence point (a completion-counter
// "last wave finishes" variant was measured SLOWER than separate finishing kernels).  So all finishing work moves
// to the CONSUMER side: a projection leaves its raw fp32 accumulator behind and the next kernel applies the
// finisher while building its own operand fragments; kernel boundaries are the only synchronisation.
//   XIN_BF16        operand = bf16 activations (o_proj after attention)
//   XIN_RESID_NORM  operand = bf16(w[k] * xnew[r][k]),  xnew = x_in + bf16round(pending accumulator)
//                   (residual add of the previous projection + the elementwise half of RMSNorm; the per-row
//                   rsqrt(mean(xnew^2)) factor is a scalar of the output row, so it is applied by the consumer of
//                   THIS projection's accumulator).  The waves of weight group 0 also write xnew to x_out and add
//                   their slab's sum of squares to ss_out[r].
//   XIN_SWIGLU      operand = bf16( bf16(silu(g)) * u ),  g, u = bf16(rstd[r] * gate/up accumulator)
// Every launch also clears up to two accumulators that earlier kernels have fully consumed and one 32-float
// statistics slot (spread over all waves), so the captured graph carries no memset nodes.
enum { XIN_BF16 = 0, XIN_RESID_NORM = 1, XIN_SWIGLU = 2 };

struct DecodeIn {
  const float* x_in; const float* pend; int64_t ld_pend; const float* norm_w; float* x_out; float* ss_out;   // RESID_NORM
  const float* gu; int64_t ld_gu; const float* ss_in; float eps; int norm_cols;                              // SWIGLU
  float* zero0; float* zero1; float* ss_zero;                                                               // clears
  int n0_4, per0, n1_4, per1;           // float4 counts and every workgroup's share of them (set_clear_shares, on the host)
};

// Round 5: everything uniform is 32-bit and comes from the host.  The round-4 form divided int64 counts by the grid size in every
// workgroup -- ~150 dependent scalar instructions per division, twice, AHEAD of the kernel's first load (the compiler hoists them
// over the early-exit of surplus workgroups): 540-600 instructions and 1.5-2 us between a wave's first instruction and its first
// load (profiles/r05_ar_prefetch.md, section 3).
__device__ __forceinline__ void decode_clear(const DecodeIn& f, int tid, int bid) {
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  const int nt = blockDim.x;
  if (f.zero0) {
    const int lo = bid * f.per0, hi = min(f.n0_4, lo + f.per0);
    for (int i = lo + tid; i < hi; i += nt) reinterpret_cast<float4*>(f.zero0)[i] = z;
  }
  if (f.zero1) {
    const int lo = bid * f.per1, hi = min(f.n1_4, lo + f.per1);
    for (int i = lo + tid; i < hi; i += nt) reinterpret_cast<float4*>(f.zero1)[i] = z;
  }
  if (f.ss_zero && bid == 0 && tid < 32) f.ss_zero[tid] = 0.f;
}

// linear workgroup id (x fastest: the order the dispatcher deals workgroups out to the XCDs in)
__device__ __forceinline__ int linear_block() { return (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x; }

__device__ __forceinline__ float silu_bf(float g) { return bf2f(f2bf(g / (1.f + __expf(-g)))); }

// Request this lane's operand pieces (row `ar`, 8 k-values at k0) for the fp32-operand modes -- hand-issued loads (vmem_asm.h): the
// values may be used only behind the caller's wait + tie.
template <int XIN>
__device__ __forceinline__ void decode_operand_load(const float* p0, const float* p1, int ld1, const float* nw, int ar, int k0, int K,
                                                    f32x4_t (&a)[2], f32x4_t (&b)[2], f32x4_t (&w)[2]) {
  if constexpr (XIN == XIN_RESID_NORM) {          // p0 = x_in [.][K], p1 = pending accumulator [.][ld1], nw = norm weight
    const float* xp = p0 + (__umul24(ar, K) + k0);
    const float* pp = p1 + (__umul24(ar, ld1) + k0);
    ld16(a[0], xp); ld16(a[1], xp + 4);
    ld16(b[0], pp); ld16(b[1], pp + 4);
    ld16(w[0], nw + k0); ld16(w[1], nw + k0 + 4);
  } else {                                        // p0 = gate / up accumulator [.][ld1]
    const float* gp = p0 + (__umul24(ar, ld1) + k0);
    ld16(a[0], gp); ld16(a[1], gp + 4);
    ld16(b[0], gp + K); ld16(b[1], gp + K + 4);
  }
}

template <int XIN>
__device__ __forceinline__ bf16x8_t decode_operand_make(const DecodeIn& f, f32x4_t (&a)[2], const f32x4_t (&b)[2], const f32x4_t (&w)[2],
                                                        float rs, float& ssq) {
  const float av[8] = {a[0][0], a[0][1], a[0][2], a[0][3], a[1][0], a[1][1], a[1][2], a[1][3]};
  const float bv[8] = {b[0][0], b[0][1], b[0][2], b[0][3], b[1][0], b[1][1], b[1][2], b[1][3]};
  bf16x8_t o;
  if constexpr (XIN == XIN_RESID_NORM) {
    const float wv[8] = {w[0][0], w[0][1], w[0][2], w[0][3], w[1][0], w[1][1], w[1][2], w[1][3]};
    float xn[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      xn[e] = av[e] + bf2f(f2bf(bv[e]));
      o[e] = (short)f2bf(wv[e] * xn[e]);
      ssq += xn[e] * xn[e];
    }
    // the updated residual values stay in `a`: the workgroups of weight group 0 store them at the kernel's END (a store issued here
    // would sit between the weight tiles' loads in the memory queue)
    a[0] = f32x4_t{xn[0], xn[1], xn[2], xn[3]};
    a[1] = f32x4_t{xn[4], xn[5], xn[6], xn[7]};
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (short)f2bf(silu_bf(bf2f(f2bf(rs * av[e]))) * bf2f(f2bf(rs * bv[e])));
  }
  return o;
}

// Addressing shared by the two ring kernels (round 5: 32-bit, 24-bit multiplies at full VALU rate; the host checks the ranges).
// A weight tile = 16 rows x 256 k of W [N][K] (row stride ldw elements); DMA instruction i of a tile covers rows 2i, 2i + 1 x 512
// bytes: lane -> row 2i + (lane >> 5), 16-byte chunk (lane & 31) ^ row (the bank swizzle, applied on the SOURCE side).
struct TileAddr {
  uint32_t lane_off[8];                          // BYTE offset of this lane inside a tile for DMA instruction i (row clamp aside)
  __device__ __forceinline__ void init(int lane, int kbase, int K, int ldw) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = 2 * i + (lane >> 5);
      lane_off[i] = (uint32_t)(__umul24(r, ldw) + min(kbase + ((lane & 31) ^ r) * 8, K - 8)) * 2u;       // chunks past K are never consumed
    }
  }
  // N % 16 != 0: the last group's missing rows re-read row N - 1 (their products are never stored)
  static __device__ __forceinline__ uint32_t ragged(int lane, int row0, int i, int kbase, int N, int K, int ldw) {
    const int r = 2 * i + (lane >> 5);
    return (uint32_t)(__umul24(min(row0 + r, N - 1), ldw) + min(kbase + ((lane & 31) ^ r) * 8, K - 8)) * 2u;
  }
};

// XIN_BF16: one wave per workgroup (see above).  Grid (k-slabs, groups of KW weight-row groups).
template <int RB, int KW>
__global__ __launch_bounds__(64) void gemv_ring_kernel(const bf16_t* __restrict__ x, int ldx, int R, const bf16_t* __restrict__ W,
                                                       int ldw, float* __restrict__ acc, int sr, int sn, int N, int K, DecodeIn f) {
  __shared__ __attribute__((aligned(1024))) char tile[2][8192];
  const int lane = threadIdx.x, g = lane >> 4, row = lane & 15;
  const int grp0 = blockIdx.y * KW;
  const int kbase = blockIdx.x * 256;
  const bool whole = (N & 15) == 0;                                 // every 16-row group is complete: no per-row clamp
  // the clears this launch carries go out first: stores behind the loads would sit between them in the memory queue
  decode_clear(f, lane, linear_block());
  TileAddr ta;
  ta.init(lane, kbase, K, ldw);
  const uint32_t lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(tile[0]));
  auto stage = [&](int t) {
    // nt: every weight byte is read once per step by one CU; leaving it out of the caches shortens issue -> landed (guide, row nt-weights)
    const uint32_t dst = lds0 + (t & 1) * 8192;
    if (whole) {
      const uint64_t base = (uint64_t)(W + (int64_t)min((grp0 + t) * 16, N - 16) * ldw);
#pragma unroll
      for (int i = 0; i < 8; ++i) dma16_nt(base, ta.lane_off[i], dst + i * 1024);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) dma16_nt((uint64_t)W, ta.ragged(lane, (grp0 + t) * 16, i, kbase, N, K, ldw), dst + i * 1024);
    }
  };
  // the activation fragments' loads go out AHEAD of the weight DMA (round 5, same-box A/B: +0.8 % tokens/s; the other order queues
  // them behind 8-16 tile instructions in the CU's vector-memory pipeline, and the MFMAs need both)
  bf16x8_t xf[RB][8];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const bf16_t* xp = x + (__umul24(min(rb * 16 + row, R - 1), ldx) + g * 8);
#pragma unroll
    for (int u = 0; u < 8; ++u) ld16(xf[rb][u], xp + min(kbase + u * 32, K - 32));
  }
  stage(0);
  if constexpr (KW > 1) stage(1);
#pragma unroll
  for (int t = 0; t < KW; ++t) {
    if (t + 1 < KW) wait_vm<8>(); else wait_vm<0>();               // loads issued behind tile t: tile t + 1's eight
    if (t == 0) {
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int u = 0; u < 8; ++u) tie(xf[rb][u]);
    }
    const char* tr = tile[t & 1] + row * 512;
    bf16x8_t wf[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) wf[u] = *reinterpret_cast<const bf16x8_t*>(tr + (((u * 4 + g) ^ row) << 4));
    if (t + 2 < KW) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      stage(t + 2);
    }
    f32x4_t d[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) d[rb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (kbase + u * 32 < K) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) d[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[rb][u], wf[u], d[rb], 0, 0, 0);
      }
    const int n = (grp0 + t) * 16 + row;
    float* ap = acc + __umul24(n, sn);
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = rb * 16 + g * 4 + j;
        if (r < R && n < N) atomicAdd(ap + __umul24(r, sr), d[rb][j]);
      }
    }
  }
}

// fp32-operand modes: FOUR waves per workgroup share one k-slab (each owns KW weight groups and its own LDS ring).
// The operand fragments are built cooperatively -- wave w converts k-steps 2w, 2w+1 for all rows and parks them in
// LDS -- because every wave re-reading the fp32 slab itself costs more L2 bandwidth than the weights cost HBM
// (measured: gate_up 14.5 -> 28.8 us).
// Grid: (k-slabs, chunks of NW x KW weight-row groups), or -- many slabs (the down projection) -- (8, chunks, ceil(slabs / 8)):
// the workgroups that share a slab -- and therefore its fp32 operand, which the previous launch's atomics left at the device
// coherence point -- then sit on ONE XCD (workgroups go to the XCDs round robin in x-fastest linear order), so the slab comes
// through the slow path once per XCD instead of once per workgroup (7.7 -> 1.1 MB per launch); surplus workgroups of the padded
// grid (slab >= nslabs) only take part in the clears.  Either way slab = x + 8 z, chunk = y: no division in the kernel.
// Argument order: the 14 dwords in front are what the operand loads and the weight tiles need (p0 / p1 / ld1 / nw: see
// decode_operand_load; p1 = the statistics slot for XIN_SWIGLU); built with kernarg preload (Makefile) they arrive in SGPRs with the wave.
// CLR = the launch carries clears (its first memory operations, so their arguments are fetched up front); a launch without them
// fetches the rest of its arguments behind the first tiles' requests.
template <int RB, int KW, int XIN, int NW, bool CLR>
__global__ __launch_bounds__(64 * NW) void gemv_ring4_kernel(const float* __restrict__ p0, const float* __restrict__ p1, const float* __restrict__ nw,
                                                             const bf16_t* __restrict__ W, int R, int K, int ld1, int ldw, int N, int nslabs,
                                                             float* __restrict__ acc, int sr, int sn, DecodeIn f) {
  __shared__ __attribute__((aligned(1024))) char tile[NW][2][8192];
  __shared__ __attribute__((aligned(16))) bf16x8_t frag[RB][8][64];
  constexpr int UPW = (8 + NW - 1) / NW;           // k-steps of the operand each wave converts
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, row = lane & 15;
  const int slab = blockIdx.x + 8 * blockIdx.z, chunk = blockIdx.y;
  asm volatile("" ::"s"(p0), "s"(p1), "s"(nw), "s"(W), "s"(R), "s"(K), "s"(ld1), "s"(ldw), "s"(N), "s"(nslabs));      // batch 1 (vmem_asm.h, argument hoisting)
  auto rest_of_args = [&] {                        // batch 2: ONE asm statement (gridDim / blockDim are implicit arguments)
    if constexpr (CLR)
      asm volatile("" ::"s"(acc), "s"(sr), "s"(sn), "s"(gridDim.x), "s"(gridDim.y), "s"(blockDim.x), "s"(f.zero0), "s"(f.zero1), "s"(f.ss_zero),
                   "s"(f.n0_4), "s"(f.per0), "s"(f.n1_4), "s"(f.per1), "s"(f.x_out), "s"(f.ss_out), "s"(__float_as_int(f.eps)), "s"(f.norm_cols));
    else
      asm volatile("" ::"s"(acc), "s"(sr), "s"(sn), "s"(f.x_out), "s"(f.ss_out), "s"(__float_as_int(f.eps)), "s"(f.norm_cols));
  };
  if constexpr (CLR) {
    rest_of_args();
    decode_clear(f, threadIdx.x, linear_block());  // the clears this launch carries go out first (see gemv_ring_kernel)
  }
  if (slab >= nslabs) return;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int grp0 = (chunk * NW + wave_u) * KW;
  const int kbase = slab * 256;
  const bool whole = (N & 15) == 0;
  // Every load of this kernel is hand-issued (vmem_asm.h).  With the builtins the operand image below -- an LDS write -- was preceded by
  // a compiler-inserted s_waitcnt vmcnt(0): it was built only after BOTH weight tiles had landed (5-6 us into the gate/up launch), and
  // the ring's third tile could only be requested after that.  Now the image is built as soon as the operand's own round trip ends.
  // operand loads of this wave's k-steps go out first, then the weight DMA
  f32x4_t a[RB][UPW][2], b[RB][UPW][2], w[RB][UPW][2];
  float rs[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const int ar = min(rb * 16 + row, R - 1);
    rs[rb] = 0.f;
    if constexpr (XIN == XIN_SWIGLU) ld4(rs[rb], p1 + ar);
#pragma unroll
    for (int uu = 0; uu < UPW; ++uu) {
      const int u = min(wave + uu * NW, 7);
      decode_operand_load<XIN>(p0, p1, ld1, nw, ar, min(kbase + u * 32, K - 32) + g * 8, K, a[rb][uu], b[rb][uu], w[rb][uu]);
    }
  }
  TileAddr ta;
  ta.init(lane, kbase, K, ldw);
  const uint32_t lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(tile[wave][0]));
  auto stage = [&](int t) {
    const uint32_t dst = lds0 + (t & 1) * 8192;
    if (whole) {
      const uint64_t base = (uint64_t)(W + (int64_t)min((grp0 + t) * 16, N - 16) * ldw);
#pragma unroll
      for (int i = 0; i < 8; ++i) dma16_nt(base, ta.lane_off[i], dst + i * 1024);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) dma16_nt((uint64_t)W, ta.ragged(lane, (grp0 + t) * 16, i, kbase, N, K, ldw), dst + i * 1024);
    }
  };
  stage(0);
  if constexpr (KW > 1) stage(1);
  if constexpr (!CLR) rest_of_args();               // (their round trip hides under the operand's)
  wait_vm<(KW > 1 ? 16 : 8)>();                     // the operand's loads are older than the tiles' DMA
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    if constexpr (XIN == XIN_SWIGLU) tie(rs[rb]);
#pragma unroll
    for (int uu = 0; uu < UPW; ++uu) {
      tie(a[rb][uu][0]); tie(a[rb][uu][1]); tie(b[rb][uu][0]); tie(b[rb][uu][1]);
      if constexpr (XIN == XIN_RESID_NORM) { tie(w[rb][uu][0]); tie(w[rb][uu][1]); }
    }
  }
  float ssq_row[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    if constexpr (XIN == XIN_SWIGLU) rs[rb] = rsqrtf(rs[rb] / (float)f.norm_cols + f.eps);
    float ssq = 0.f;
#pragma unroll
    for (int uu = 0; uu < UPW; ++uu) {
      const int u = wave + uu * NW;
      if (u < 8) {
        const bool in_k = kbase + u * 32 < K;                        // clamped (re-read) steps carry no new data
        float part = 0.f;
        frag[rb][u][lane] = decode_operand_make<XIN>(f, a[rb][uu], b[rb][uu], w[rb][uu], rs[rb], part);
        if (in_k) ssq += part;
      }
    }
    if constexpr (XIN == XIN_RESID_NORM) {
      ssq += __shfl_xor(ssq, 16, 64);
      ssq += __shfl_xor(ssq, 32, 64);
    }
    ssq_row[rb] = ssq;
  }
  lds_barrier();
  bf16x8_t xf[RB][8];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int u = 0; u < 8; ++u) xf[rb][u] = frag[rb][u][lane];
#pragma unroll
  for (int t = 0; t < KW; ++t) {
    if (t + 1 < KW) wait_vm<8>(); else wait_vm<0>();               // loads issued behind tile t: tile t + 1's eight
    const char* tr = tile[wave][t & 1] + row * 512;
    bf16x8_t wf[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) wf[u] = *reinterpret_cast<const bf16x8_t*>(tr + (((u * 4 + g) ^ row) << 4));
    if (t + 2 < KW) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      stage(t + 2);
    }
    f32x4_t d[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) d[rb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (kbase + u * 32 < K) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) d[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[rb][u], wf[u], d[rb], 0, 0, 0);
      }
    const int n = (grp0 + t) * 16 + row;
    float* ap = acc + __umul24(n, sn);
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = rb * 16 + g * 4 + j;
        if (r < R && n < N) atomicAdd(ap + __umul24(r, sr), d[rb][j]);
      }
    }
  }
  // the updated residual stream and its row statistics (weight group 0's workgroups) leave last: no store sits between the loads above
  if constexpr (XIN == XIN_RESID_NORM) {
    if (chunk == 0) {
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        const bool live = rb * 16 + row < R;
#pragma unroll
        for (int uu = 0; uu < UPW; ++uu) {
          const int u = wave + uu * NW;
          if (u < 8 && live && kbase + u * 32 < K) {
            float* op = f.x_out + (__umul24(rb * 16 + row, K) + kbase + u * 32 + g * 8);
            *reinterpret_cast<f32x4_t*>(op) = a[rb][uu][0];
            *reinterpret_cast<f32x4_t*>(op + 4) = a[rb][uu][1];
          }
        }
        if (live && g == 0 && f.ss_out && ssq_row[rb] != 0.f) atomicAdd(f.ss_out + rb * 16 + row, ssq_row[rb]);
      }
    }
  }
}


template <int RB>
void launch_gemv(int U, dim3 grid, hipStream_t st, const bf16_t* x, int64_t ldx, int R, const bf16_t* W, int64_t ldw,
                 float* acc, int64_t sr, int64_t sn, int N, int K) {
  const dim3 block(256);
  switch (U) {
    case 8: hipLaunchKernelGGL((gemv_kernel<RB, 8>), grid, block, 0, st, x, ldx, R, W, ldw, acc, sr, sn, N, K); break;
    case 4: hipLaunchKernelGGL((gemv_kernel<RB, 4>), grid, block, 0, st, x, ldx, R, W, ldw, acc, sr, sn, N, K); break;
    case 2: hipLaunchKernelGGL((gemv_kernel<RB, 2>), grid, block, 0, st, x, ldx, R, W, ldw, acc, sr, sn, N, K); break;
    default: hipLaunchKernelGGL((gemv_kernel<RB, 1>), grid, block, 0, st, x, ldx, R, W, ldw, acc, sr, sn, N, K); break;
  }
}

// every workgroup's share of the clears a launch carries (float4 units; the kernels only multiply)
void set_clear_shares(DecodeIn& f, float* zero0, int64_t n0, float* zero1, int64_t n1, float* ss_zero, unsigned nblocks) {
  f.zero0 = zero0; f.zero1 = zero1; f.ss_zero = ss_zero;
  f.n0_4 = (int)(n0 >> 2); f.n1_4 = (int)(n1 >> 2);
  f.per0 = (int)((f.n0_4 + nblocks - 1) / nblocks); f.per1 = (int)((f.n1_4 + nblocks - 1) / nblocks);
}

// measured on MI355X (tools/gemv_bench.py): two row groups per wave once there are >= 3000 tiles, else one
template <int XIN>
void launch_ring_auto(hipStream_t st, const bf16_t* x, int ldx, int R, const bf16_t* W, int ldw, float* acc, int sr, int sn, int N,
                      int K, DecodeIn f, float* zero0 = nullptr, int64_t n0 = 0, float* zero1 = nullptr, int64_t n1 = 0,
                      float* ss_zero = nullptr) {
  const int64_t groups = (N + 15) / 16;
  const int nslabs = (K + 255) / 256;
  const int KW = groups * nslabs >= 3000 ? 2 : 1;
  if constexpr (XIN == XIN_BF16) {
    const dim3 grid((unsigned)nslabs, (unsigned)((groups + KW - 1) / KW));
    set_clear_shares(f, zero0, n0, zero1, n1, ss_zero, grid.x * grid.y);
#define UG_RING1(RBV, KWV) hipLaunchKernelGGL((gemv_ring_kernel<RBV, KWV>), grid, dim3(64), 0, st, x, ldx, R, W, ldw, acc, sr, sn, N, K, f)
    if (R <= 16) { if (KW == 2) UG_RING1(1, 2); else UG_RING1(1, 1); }
    else { if (KW == 2) UG_RING1(2, 2); else UG_RING1(2, 1); }
#undef UG_RING1
  } else {
    // workgroup shape (waves, tiles per wave) by the tile count the busiest CU ends up with: one 9-wave x 3-tile workgroup
    // per CU for gate_up (27 tile slots for 26.25 tiles per CU), 7 x 2 for down (14 for 13.1), 4 x 1|2 otherwise
    struct Cand { int nw, kw; };
    const Cand cands[4] = {{4, 1}, {4, 2}, {7, 2}, {9, 3}};
    int best = 0;
    int64_t best_cost = INT64_MAX;
    for (int c = 0; c < 4; ++c) {
      if (R > 16 && cands[c].nw != 4) continue;                    // two row blocks: the operand image needs the LDS
      const int64_t blocks = ((groups + cands[c].nw * cands[c].kw - 1) / (cands[c].nw * cands[c].kw)) * nslabs;
      const int64_t cost = ((blocks + 255) / 256) * cands[c].nw * cands[c].kw;
      if (cost < best_cost) { best_cost = cost; best = c; }
    }
    int NWc = cands[best].nw, KWc = cands[best].kw, xcd_chunks = 0;
    // many k-slabs (down projection: 35): the workgroups of a slab on one XCD -- 8 waves x 2 tiles, chunks x ceil(slabs / 8)
    // workgroups per XCD must fit its 32 CUs
    static const int xcd_major = [] { const char* e = getenv("UNIGEN_DECODE_XCD_SLABS"); return e ? atoi(e) : 1; }();
    if (xcd_major && R <= 16 && nslabs >= 16) {
      const int64_t chunks8 = (groups + 15) / 16;
      if (chunks8 * ((nslabs + 7) / 8) <= 32) { NWc = 8; KWc = 2; xcd_chunks = (int)chunks8; }
    }
    // (dealing the nslabs % 8 left-over slabs' workgroups over all XCDs for equal bytes per XCD measured no better: 10.50 vs 10.33 us)
    const dim3 grid = xcd_chunks ? dim3(8, (unsigned)xcd_chunks, (unsigned)((nslabs + 7) / 8))
                                 : dim3((unsigned)nslabs, (unsigned)((groups + NWc * KWc - 1) / (NWc * KWc)));
    set_clear_shares(f, zero0, n0, zero1, n1, ss_zero, grid.x * grid.y * grid.z);
    // leading arguments: see the kernel
    const float* p0 = XIN == XIN_RESID_NORM ? f.x_in : f.gu;
    const float* p1 = XIN == XIN_RESID_NORM ? f.pend : f.ss_in;
    const int64_t ld1 = XIN == XIN_RESID_NORM ? f.ld_pend : f.ld_gu;          // (< 2^24: checked by the entry points)
    const bool clr = zero0 || zero1 || ss_zero;
#define UG_RING4(RBV, KWV, NWV)                                                                                                              \
  do {                                                                                                                                       \
    if (clr) hipLaunchKernelGGL((gemv_ring4_kernel<RBV, KWV, XIN, NWV, true>), grid, dim3(64 * NWV), 0, st, p0, p1, f.norm_w, W, R, K, (int)ld1, ldw, \
                                N, nslabs, acc, sr, sn, f);                                                                                  \
    else hipLaunchKernelGGL((gemv_ring4_kernel<RBV, KWV, XIN, NWV, false>), grid, dim3(64 * NWV), 0, st, p0, p1, f.norm_w, W, R, K, (int)ld1, ldw, \
                            N, nslabs, acc, sr, sn, f);                                                                                      \
  } while (0)
    if (R <= 16) {
      if (NWc == 9) UG_RING4(1, 3, 9); else if (NWc == 8) UG_RING4(1, 2, 8); else if (NWc == 7) UG_RING4(1, 2, 7); else if (KWc == 2) UG_RING4(1, 2, 4); else UG_RING4(1, 1, 4);
    } else { if (KWc == 2) UG_RING4(2, 2, 4); else UG_RING4(2, 1, 4); }
#undef UG_RING4
  }
}

// 32-bit / 24-bit ranges of the ring kernels' addressing (weights up to 2^31 elements, strides below 2^24)
// ... and of their grids: weight-row groups ride in grid.y (at most 65 535 workgroups: N <= 16 x 65 535 rows with one group per
// workgroup, the smallest unit any launch shape uses), k-slabs in grid.x or, eight at a time, in grid.z
#define UG_RING_RANGES(name, N, K, ldw, sr, sn)                                                                                      \
  UG_REQUIRE((N) < (1 << 24) && (K) < (1 << 24) && (ldw) < (1 << 24) && (int64_t)(N) * (ldw) < (1ll << 31) && (sr) < (1 << 24) &&    \
                 (sn) < (1 << 24) && (int64_t)(N) * (sn) + 32 * (int64_t)(sr) < (1ll << 31),                                         \
             name ": sizes beyond the kernels' 32-bit addressing (N=%ld K=%ld ldw=%ld)", (long)(N), (long)(K), (long)(ldw));         \
  UG_REQUIRE(((N) + 15) / 16 <= 65535 && ((K) + 255) / 256 <= 65535 * 8,                                                              \
             name ": N = %ld weight rows / K = %ld exceed the launch grid (16 x 65 535 rows, 2 048 x 65 535 columns)", (long)(N), (long)(K))

}  // namespace

extern "C" int ug_gemv_bf16(const void* x, int64_t ldx, int64_t R, const void* W, int64_t ldw, float* acc, int64_t acc_stride_r,
                            int64_t acc_stride_n, int64_t N, int64_t K, hipStream_t st) {
  UG_REQUIRE(R > 0 && R <= 32 && N > 0 && K > 0 && K % 32 == 0, "ug_gemv_bf16: need 1 <= rows <= 32 and K %% 32 == 0 (rows=%ld K=%ld)", (long)R, (long)K);
  UG_REQUIRE(ldx % 8 == 0 && ldw % 8 == 0 && ug_aligned16(x) && ug_aligned16(W), "ug_gemv_bf16: 16-byte aligned rows required");
  const int64_t groups = (N + 15) / 16;
  const bf16_t* xb = (const bf16_t*)x;
  const bf16_t* wb = (const bf16_t*)W;
  if (K >= 256) {
    UG_RING_RANGES("ug_gemv_bf16", N, K, ldw, acc_stride_r, acc_stride_n);
    UG_REQUIRE(ldx < (1 << 24), "ug_gemv_bf16: ldx beyond 2^24");
    launch_ring_auto<XIN_BF16>(st, xb, (int)ldx, (int)R, wb, (int)ldw, acc, (int)acc_stride_r, (int)acc_stride_n, (int)N, (int)K, DecodeIn{});
    UG_CHECK_LAUNCH("ug_gemv_bf16");
    return UG_OK;
  }
  // short contractions: direct loads, deepest per-wave unroll that still leaves >= 2048 waves
  int U = 8;
  while (U > 1 && groups * ((K + 32 * U - 1) / (32 * U)) < 2048) U >>= 1;
  const int64_t nslices = (K + 32 * U - 1) / (32 * U);
  dim3 grid((unsigned)groups, (unsigned)((nslices + 3) / 4));
  if (R <= 16) launch_gemv<1>(U, grid, st, xb, ldx, (int)R, wb, ldw, acc, acc_stride_r, acc_stride_n, (int)N, (int)K);
  else launch_gemv<2>(U, grid, st, xb, ldx, (int)R, wb, ldw, acc, acc_stride_r, acc_stride_n, (int)N, (int)K);
  UG_CHECK_LAUNCH("ug_gemv_bf16");
  return UG_OK;
}

#define UG_DECODE_COMMON(name)                                                                                              \
  UG_REQUIRE(R > 0 && R <= 32 && K >= 256 && K % 32 == 0 && N > 0 && ldw % 8 == 0 && ug_aligned16(W) && acc &&              \
                 ug_aligned16(acc) && ldacc % 4 == 0 && n0 % 4 == 0 && n1 % 4 == 0 && ug_aligned16(zero0) && ug_aligned16(zero1), \
             name ": need 1 <= rows <= 32, K >= 256, K %% 32 == 0, 16-byte aligned operands (rows=%ld K=%ld)", (long)R, (long)K);                        \
  UG_RING_RANGES(name, N, K, ldw, ldacc, 1);                                                                                 \
  UG_REQUIRE(n0 < (1ll << 31) && n1 < (1ll << 31), name ": clears beyond 2^31 floats")

extern "C" int ug_decode_gemv(const void* x, int64_t ldx, int64_t R, const void* W, int64_t ldw, float* acc, int64_t ldacc,
                              int64_t N, int64_t K, float* zero0, int64_t n0, float* zero1, int64_t n1, float* ss_zero,
                              hipStream_t st) {
  UG_DECODE_COMMON("ug_decode_gemv");
  UG_REQUIRE(x && ldx % 8 == 0 && ug_aligned16(x), "ug_decode_gemv: activations must be 16-byte aligned rows");
  UG_REQUIRE(ldx < (1 << 24), "ug_decode_gemv: ldx beyond 2^24");
  launch_ring_auto<XIN_BF16>(st, (const bf16_t*)x, (int)ldx, (int)R, (const bf16_t*)W, (int)ldw, acc, (int)ldacc, 1, (int)N, (int)K, DecodeIn{},
                             zero0, n0, zero1, n1, ss_zero);
  UG_CHECK_LAUNCH("ug_decode_gemv");
  return UG_OK;
}

extern "C" int ug_decode_gemv_resid_norm(const float* x_in, const float* pending, int64_t ld_pending, const float* norm_w,
                                         float* x_out, float* ss_out, int64_t R, const void* W, int64_t ldw, float* acc,
                                         int64_t ldacc, int64_t N, int64_t K, float* zero0, int64_t n0, float* zero1, int64_t n1,
                                         float* ss_zero, hipStream_t st) {
  UG_DECODE_COMMON("ug_decode_gemv_resid_norm");
  UG_REQUIRE(x_in && pending && norm_w && x_out && ss_out && x_in != x_out && ld_pending % 4 == 0 && ug_aligned16(x_in) &&
                 ug_aligned16(pending) && ug_aligned16(norm_w) && ug_aligned16(x_out),
             "ug_decode_gemv_resid_norm: bad args (x_in and x_out must be distinct, 16-byte aligned fp32 buffers)");
  DecodeIn f{};
  f.x_in = x_in; f.pend = pending; f.ld_pend = ld_pending; f.norm_w = norm_w; f.x_out = x_out; f.ss_out = ss_out;
  UG_REQUIRE(ld_pending < (1 << 24), "ug_decode_gemv_resid_norm: ld_pending beyond 2^24");
  launch_ring_auto<XIN_RESID_NORM>(st, nullptr, 0, (int)R, (const bf16_t*)W, (int)ldw, acc, (int)ldacc, 1, (int)N, (int)K, f, zero0, n0, zero1, n1, ss_zero);
  UG_CHECK_LAUNCH("ug_decode_gemv_resid_norm");
  return UG_OK;
}

extern "C" int ug_decode_gemv_swiglu(const float* gate_up_acc, int64_t ld_gu, const float* ss_in, float eps, int64_t norm_cols,
                                     int64_t R, const void* W, int64_t ldw, float* acc, int64_t ldacc, int64_t N, int64_t K,
                                     float* zero0, int64_t n0, float* zero1, int64_t n1, float* ss_zero, hipStream_t st) {
  UG_DECODE_COMMON("ug_decode_gemv_swiglu");
  UG_REQUIRE(gate_up_acc && ss_in && norm_cols > 0 && ld_gu % 4 == 0 && K % 4 == 0 && ug_aligned16(gate_up_acc),
             "ug_decode_gemv_swiglu: bad args");
  DecodeIn f{};
  f.gu = gate_up_acc; f.ld_gu = ld_gu; f.ss_in = ss_in; f.eps = eps; f.norm_cols = (int)norm_cols;
  UG_REQUIRE(ld_gu < (1 << 24), "ug_decode_gemv_swiglu: ld_gu beyond 2^24");
  launch_ring_auto<XIN_SWIGLU>(st, nullptr, 0, (int)R, (const bf16_t*)W, (int)ldw, acc, (int)ldacc, 1, (int)N, (int)K, f, zero0, n0, zero1, n1, ss_zero);
  UG_CHECK_LAUNCH("ug_decode_gemv_swiglu");
  return UG_OK;
}

extern "C" int ug_decode_finish_resid_norm(float* acc, int64_t ldacc, float* x, const float* w, void* xn, int64_t rows,
                                           int64_t cols, float eps, int* pos_inc, int* len_inc, hipStream_t st) {
  UG_REQUIRE((pos_inc == nullptr) == (len_inc == nullptr), "ug_decode_finish_resid_norm: pos_inc / len_inc come together");
  UG_REQUIRE(rows > 0 && cols > 0 && cols % 4 == 0 && cols <= 4096 && ldacc % 4 == 0 && acc && x && w && xn,
             "ug_decode_finish_resid_norm: bad args (cols=%ld, multiple of 4 and <= 4096)", (long)cols);
  UG_REQUIRE(ug_aligned16(x) && ug_aligned16(w) && ug_aligned16(acc) && ((uintptr_t)xn & 7) == 0,
             "ug_decode_finish_resid_norm: alignment");
  if (cols <= 2048)
    hipLaunchKernelGGL(finish_resid_norm_kernel<8>, dim3((unsigned)rows), dim3(64), 0, st, acc, ldacc, x, w, (bf16_t*)xn,
                       (int)cols, eps, pos_inc, len_inc);
  else
    hipLaunchKernelGGL(finish_resid_norm_kernel<16>, dim3((unsigned)rows), dim3(64), 0, st, acc, ldacc, x, w, (bf16_t*)xn,
                       (int)cols, eps, pos_inc, len_inc);
  UG_CHECK_LAUNCH("ug_decode_finish_resid_norm");
  return UG_OK;
}

extern "C" int ug_kv_store(const void* qkv, int64_t ldq, int64_t k_col, int64_t v_col, void* cache_k, void* cache_v,
                           int64_t rows, int64_t L, int HKV, int head_dim, int64_t Tmax, const int* pos_dev, int pos_host,
                           hipStream_t st) {
  UG_REQUIRE(rows > 0 && L > 0 && head_dim == DHD && ldq % 8 == 0 && k_col % 8 == 0 && v_col % 8 == 0, "ug_kv_store: bad args");
  UG_REQUIRE(ug_aligned16(qkv) && ug_aligned16(cache_k) && ug_aligned16(cache_v), "ug_kv_store: alignment");
  const int64_t total = rows * L * HKV * (DHD / 8);
  int64_t g = (total + 255) / 256; if (g > 2048) g = 2048;
  hipLaunchKernelGGL(kv_store_kernel, dim3((unsigned)g), dim3(256), 0, st, (const bf16_t*)qkv, ldq, (int)k_col, (int)v_col,
                     (bf16_t*)cache_k, (bf16_t*)cache_v, (int)rows, (int)L, HKV, (int)Tmax, pos_dev, pos_host);
  UG_CHECK_LAUNCH("ug_kv_store");
  return UG_OK;
}

extern "C" int ug_rope_at(void* qkv, const float* cos_tab, const float* sin_tab, int64_t rows, int64_t ldq, int nheads,
                          int head_dim, const int* pos_dev, int64_t max_pos, hipStream_t st) {
  UG_REQUIRE(rows > 0 && head_dim % 8 == 0 && pos_dev, "ug_rope_at: bad args");
  const int total = (int)(rows * nheads * (head_dim / 8));
  hipLaunchKernelGGL(rope_at_kernel, dim3((total + 255) / 256), dim3(256), 0, st, (bf16_t*)qkv, cos_tab, sin_tab, (int)rows,
                     (int)ldq, nheads, head_dim, pos_dev, (int)max_pos);
  UG_CHECK_LAUNCH("ug_rope_at");
  return UG_OK;
}

extern "C" int ug_attn_decode(const void* q, int64_t ldq, const void* cache_k, const void* cache_v, const uint8_t* key_valid,
                              void* o, int64_t ldo, int64_t rows, int H, int HKV, int head_dim, int64_t Tmax,
                              const int* len_dev, float scale, hipStream_t st) {
  UG_REQUIRE(rows > 0 && head_dim == DHD && H % HKV == 0 && len_dev, "ug_attn_decode: bad args");
  hipLaunchKernelGGL(attn_decode_kernel, dim3(H, (unsigned)rows), dim3(64 * AD_WAVES), 0, st, (const bf16_t*)q, ldq, (const bf16_t*)cache_k,
                     (const bf16_t*)cache_v, key_valid, (bf16_t*)o, ldo, H, HKV, (int)Tmax, len_dev, scale);
  UG_CHECK_LAUNCH("ug_attn_decode");
  return UG_OK;
}

extern "C" int ug_attn_decode_fused(const float* acc_qkv, int64_t ldacc, const float* ss_in, float eps, int64_t norm_cols,
                                    const void* bias, const float* cos_tab, const float* sin_tab, const int* pos_dev,
                                    void* cache_k, void* cache_v, const uint8_t* key_valid, void* o, int64_t ldo, int64_t rows,
                                    int H, int HKV, int head_dim, int64_t Tmax, int64_t max_pos, float scale, hipStream_t st) {
  UG_REQUIRE(rows > 0 && head_dim == DHD && H % HKV == 0 && acc_qkv && ss_in && norm_cols > 0 && pos_dev && cache_k && cache_v && o,
             "ug_attn_decode_fused: bad args");
  // (XCD-aware placement: see the kernel)
  const dim3 grid(8u, (unsigned)(H / HKV), (unsigned)((rows * HKV + 7) / 8));
  UG_REQUIRE(ldacc > 0 && ldacc < (1ll << 31), "ug_attn_decode_fused: ldacc out of range");
  hipLaunchKernelGGL(attn_decode_fused_kernel, grid, dim3(64 * ADF_WAVES), 0, st, acc_qkv, ss_in, pos_dev, (bf16_t*)cache_k,
                     (bf16_t*)cache_v, (int)rows, HKV, H, (int)ldacc, (int)Tmax, (int)max_pos, (const bf16_t*)bias, cos_tab, sin_tab,
                     key_valid, (bf16_t*)o, ldo, eps, (int)norm_cols, scale);
  UG_CHECK_LAUNCH("ug_attn_decode_fused");
  return UG_OK;
}

extern "C" int ug_skinny_finish(const float* acc, const void* bias, void* out_bf16, float* resid, int64_t M, int64_t N,
                                int mode, hipStream_t st) {
  UG_REQUIRE(M > 0 && N > 0 && (mode == 0 ? out_bf16 != nullptr : resid != nullptr), "ug_skinny_finish: bad args");
  const int64_t total = M * N;
  int64_t g = (total + 255) / 256; if (g > 1024) g = 1024;
  hipLaunchKernelGGL(skinny_finish_kernel, dim3((unsigned)g), dim3(256), 0, st, acc, (const bf16_t*)bias, (bf16_t*)out_bf16, resid,
                     total, (int)N, mode);
  UG_CHECK_LAUNCH("ug_skinny_finish");
  return UG_OK;
}
