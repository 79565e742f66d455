// Autoregressive decode support for UniGen.t2i_generate_ar (reference models/unigen.py:457-521, which
// drives transformers' DynamicCache + SDPA one token at a time).  Static KV cache sized for the whole
// generation, position / length read from DEVICE memory so one captured hipGraph replays all 256 steps.
//   cache layout: K,V [rows][HKV][Tmax][128] bf16 (keys of one (row, kv-head) contiguous)
#include "common.h"
#include "unigen_hip.h"

namespace {

constexpr int DHD = 128;

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

// One wave per (row, query head): lane j scores key (chunk*64 + j) against q (fp32 in LDS), online softmax
// across chunks, then lanes switch to owning two output dims and accumulate p_j * V_j.
// key_valid: optional [rows][Tmax] bytes (0 = padding key, never attended), like HF's 2-D attention_mask.
__global__ __launch_bounds__(64) void attn_decode_kernel(const bf16_t* __restrict__ q, int64_t ldq, const bf16_t* __restrict__ ck,
                                                         const bf16_t* __restrict__ cv, const uint8_t* __restrict__ key_valid,
                                                         bf16_t* __restrict__ o, int64_t ldo, int H, int HKV, int Tmax,
                                                         const int* __restrict__ len_dev, float scale) {
  __shared__ float qs[DHD];
  __shared__ float ps[64];
  const int r = blockIdx.y, h = blockIdx.x, lane = threadIdx.x;
  const int hk = h / (H / HKV);
  const int len = min(*len_dev, Tmax);
  const bf16_t* qp = q + (int64_t)r * ldq + h * DHD;
  qs[lane * 2] = bf2f(qp[lane * 2]);
  qs[lane * 2 + 1] = bf2f(qp[lane * 2 + 1]);
  __syncthreads();
  const bf16_t* kb = ck + ((int64_t)r * HKV + hk) * Tmax * DHD;
  const bf16_t* vb = cv + ((int64_t)r * HKV + hk) * Tmax * DHD;
  float m = -INFINITY, l = 0.f, acc0 = 0.f, acc1 = 0.f;
  for (int t0 = 0; t0 < len; t0 += 64) {
    const int t = t0 + lane;
    float s = -INFINITY;
    if (t < len && (!key_valid || key_valid[(int64_t)r * Tmax + t])) {
      const bf16_t* kr = kb + (int64_t)t * DHD;
      float d = 0.f;
#pragma unroll
      for (int c = 0; c < DHD / 8; ++c) {
        const bf16x8_t kv = *reinterpret_cast<const bf16x8_t*>(kr + c * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) d += bf2f((bf16_t)kv[e]) * qs[c * 8 + e];
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
    __syncthreads();
    ps[lane] = bf2f(f2bf(p));              // P is rounded to bf16 before P.V like the bf16 SDPA paths
    __syncthreads();
    acc0 *= alpha; acc1 *= alpha;
    const int nvalid = min(64, len - t0);
    for (int j = 0; j < nvalid; ++j) {
      const float pj = ps[j];
      const bf16x2_t vv = *reinterpret_cast<const bf16x2_t*>(vb + (int64_t)(t0 + j) * DHD + lane * 2);
      acc0 += pj * bf2f((bf16_t)vv[0]);
      acc1 += pj * bf2f((bf16_t)vv[1]);
    }
  }
  const float inv = l > 0.f ? 1.f / l : 0.f;
  bf16_t* op = o + (int64_t)r * ldo + h * DHD + lane * 2;
  *reinterpret_cast<uint32_t*>(op) = pack_bf2(acc0 * inv, acc1 * inv);
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

}  // namespace

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
  hipLaunchKernelGGL(attn_decode_kernel, dim3(H, (unsigned)rows), dim3(64), 0, st, (const bf16_t*)q, ldq, (const bf16_t*)cache_k,
                     (const bf16_t*)cache_v, key_valid, (bf16_t*)o, ldo, H, HKV, (int)Tmax, len_dev, scale);
  UG_CHECK_LAUNCH("ug_attn_decode");
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
