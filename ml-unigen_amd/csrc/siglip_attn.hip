// Fused fp32-accurate attention for the SigLIP vision tower (reference: models/multimodal_encoder/siglip_encoder.py:196-260,
// SigLipAttention.forward: q k^T * scale -> fp32 softmax -> p v, 16 heads of 72 over 729 patch tokens, no mask).
//
// The reference materialises [B, 16, 729, 729] fp32 scores; round 1 of this build did the same with two batched fp32-MFMA
// GEMMs and a row-softmax pass (34 MB of scores per image written and read twice per layer).  Here one flash-style kernel
// keeps the scores in registers.  The tower runs in fp32 in the reference, so both contractions use the scaled two-way f16
// operand split of conv_split.hip (three products, as accurate as an fp32 accumulation, 3/16 of the fp32-MFMA cost):
//   * q, k, v share one power-of-two scale 2^e from a device scalar bounding max|qkv| (ug_amax_f32 on the projection
//     output); probabilities (<= 1) use the fixed scale 2^14;
//   * S^T = K Q^T as in attention.hip (K fragments from LDS, Q fragments split once into registers), so every lane owns
//     one query column: the online-softmax statistics are lane-local plus two shuffles, and the split probabilities feed
//     the second contraction as its B operand straight from registers;
//   * O^T += V^T P^T with the V^T fragments read from the row-major V planes by ds_read_b64_tr_b16.
// K / V tiles of 64 keys are split on the way from registers to LDS (two f16 planes each); head_dim is padded with zero
// columns to 96 for the first contraction (three k-steps of 32) and to 80 for the second (five 16-row blocks).
#include "common.h"
#include "split_f16.h"
#include "unigen_hip.h"

namespace {

constexpr int KP = 96;          // padded contraction length of Q K^T (head_dim <= 96)
constexpr int K_LD = 104;       // K plane row pitch (elements): 208 B = 13 x 16 B, sixteen consecutive rows fall on 16 distinct slots
constexpr int DP = 80;          // padded head_dim of the P V output (five blocks of 16)
constexpr int V_LD = 80;        // V plane row pitch: 160 B = 5 x 32 B, eight consecutive rows fall on 8 distinct 32-byte slots
constexpr int NKS = KP / 32, NDB = DP / 16;
constexpr int P_EXP = 14;       // probabilities are scaled by 2^14 before their split
typedef __attribute__((ext_vector_type(4))) short s16x4_t;

struct SigArgs {
  const float* qkv; float* out; const float* amax;
  int64_t ld, ldo;
  int B, T, H, hd;
  float scale;
};

__device__ __forceinline__ h16x8_t as_h8(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  const u32x4_t v = {a, b, c, d};
  return __builtin_bit_cast(h16x8_t, v);
}

// tile rows (keys) key0 .. key0+63 of a head's K or V slice -> two f16 planes [64][LD]; columns >= hd stay zero (cleared once)
template <int LD>
__device__ __forceinline__ void stage_split(bf16_t* p1, bf16_t* p2, const float* src, int64_t ld, int key0, int T, int hd4, int ex, int tid) {
  const int chunks = 64 * hd4;
  for (int c = tid; c < chunks; c += 256) {
    const int key = c / hd4, q4 = c - key * hd4;
    const int gk = min(key0 + key, T - 1);                   // rows past T repeat the last key: their probabilities are exact zeros
    const float4 v = *reinterpret_cast<const float4*>(src + (int64_t)gk * ld + q4 * 4);
    uint32_t a1, a2, b1, b2;
    split2_pair(__builtin_ldexpf(v.x, ex), __builtin_ldexpf(v.y, ex), a1, a2);
    split2_pair(__builtin_ldexpf(v.z, ex), __builtin_ldexpf(v.w, ex), b1, b2);
    *reinterpret_cast<uint2*>(p1 + key * LD + q4 * 4) = make_uint2(a1, b1);
    *reinterpret_cast<uint2*>(p2 + key * LD + q4 * 4) = make_uint2(a2, b2);
  }
}

__device__ __forceinline__ h16x8_t frag_k(const bf16_t* t, int jb, int ks, int lane) {
  return *reinterpret_cast<const h16x8_t*>(t + (jb * 16 + (lane & 15)) * K_LD + ks * 32 + (lane >> 4) * 8);
}
// transposed fragment of a row-major V plane: rows d = db*16.., contraction over keys with the k-slot order of attention.hip
// (slot s of lane-group g <-> key jp*32 + (s>>2)*16 + g*4 + (s&3))
__device__ __forceinline__ h16x8_t frag_vt(const bf16_t* t, int db, int jp, int lane) {
  const int i16 = lane & 15, g = lane >> 4;
  const bf16_t* p0 = t + (jp * 32 + g * 4 + (i16 >> 2)) * V_LD + db * 16 + (i16 & 3) * 4;
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p0);
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p0 + 16 * V_LD));
  return __builtin_bit_cast(h16x8_t, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
__device__ __forceinline__ float group_max(float v) { v = fmaxf(v, __shfl_xor(v, 16, 64)); return fmaxf(v, __shfl_xor(v, 32, 64)); }
__device__ __forceinline__ float group_sum(float v) { v += __shfl_xor(v, 16, 64); return v + __shfl_xor(v, 32, 64); }

// grid (ceil(T / 64), H, B); 4 waves x 16 query rows
__global__ __launch_bounds__(256, 2) void siglip_attn_kernel(SigArgs p) {
  __shared__ __attribute__((aligned(16))) bf16_t K1[64 * K_LD];
  __shared__ __attribute__((aligned(16))) bf16_t K2[64 * K_LD];
  __shared__ __attribute__((aligned(16))) bf16_t V1[64 * V_LD];
  __shared__ __attribute__((aligned(16))) bf16_t V2[64 * V_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
  const int qt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int D = p.H * p.hd, hd4 = p.hd >> 2;
  const int ex = scale_exp(p.amax);
  const float s_unscale = __builtin_ldexpf(p.scale, -2 * ex);          // scores: acc * 2^-(eq+ek) * softmax scale
  const float o_unscale = __builtin_ldexpf(1.f, -(ex + P_EXP));
  const float* qbase = p.qkv + (int64_t)b * p.T * p.ld + h * p.hd;
  const float* kbase = qbase + D;
  const float* vbase = qbase + 2 * D;

  for (int i = tid; i < 64 * K_LD / 2; i += 256) { reinterpret_cast<uint32_t*>(K1)[i] = 0u; reinterpret_cast<uint32_t*>(K2)[i] = 0u; }
  for (int i = tid; i < 64 * V_LD / 2; i += 256) { reinterpret_cast<uint32_t*>(V1)[i] = 0u; reinterpret_cast<uint32_t*>(V2)[i] = 0u; }

  // this lane's query row, split once: B operand of the first contraction (k = ks*32 + 8g .. +7)
  const int qrow = qt * 64 + wave * 16 + (lane & 15);
  const int qrow_c = min(qrow, p.T - 1);
  h16x8_t q1[NKS], q2[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    uint32_t w1[4], w2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int col = ks * 32 + g * 8 + e * 2;
      float a = 0.f, c = 0.f;
      if (col < p.hd) {                                      // hd % 2 == 0: a pair is inside or outside as a whole
        const float2 v = *reinterpret_cast<const float2*>(qbase + (int64_t)qrow_c * p.ld + col);
        a = __builtin_ldexpf(v.x, ex); c = __builtin_ldexpf(v.y, ex);
      }
      split2_pair(a, c, w1[e], w2[e]);
    }
    q1[ks] = as_h8(w1[0], w1[1], w1[2], w1[3]);
    q2[ks] = as_h8(w2[0], w2[1], w2[2], w2[3]);
  }

  f32x4_t ot[NDB];
#pragma unroll
  for (int d = 0; d < NDB; ++d) ot[d] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float m_i = -INFINITY, l_i = 0.f;
  const int ntile = (p.T + 63) / 64;

  for (int t = 0; t < ntile; ++t) {
    __syncthreads();
    stage_split<K_LD>(K1, K2, kbase, p.ld, t * 64, p.T, hd4, ex, tid);
    stage_split<V_LD>(V1, V2, vbase, p.ld, t * 64, p.T, hd4, ex, tid);
    __syncthreads();

    f32x4_t st[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      st[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const h16x8_t k1 = frag_k(K1, j, ks, lane), k2 = frag_k(K2, j, ks, lane);
        st[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(k1, q2[ks], st[j], 0, 0, 0);
        st[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(k2, q1[ks], st[j], 0, 0, 0);
        st[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(k1, q1[ks], st[j], 0, 0, 0);
      }
    }
    float mloc = -INFINITY;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool on = t * 64 + j * 16 + g * 4 + r < p.T;
        const float s = on ? st[j][r] * s_unscale : -INFINITY;
        st[j][r] = s;
        mloc = fmaxf(mloc, s);
      }
    mloc = group_max(mloc);
    const float m_new = fmaxf(m_i, mloc);
    const float alpha = __expf(m_i - m_new);                 // every tile holds at least one key, so m_new is finite
    float rs = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float e = __expf(st[j][r] - m_new); st[j][r] = e; rs += e; }
    rs = group_sum(rs);
    l_i = l_i * alpha + rs;
    m_i = m_new;
#pragma unroll
    for (int d = 0; d < NDB; ++d) { ot[d][0] *= alpha; ot[d][1] *= alpha; ot[d][2] *= alpha; ot[d][3] *= alpha; }
    // probabilities -> two f16 planes in the B-operand layout (slots 0-3: block 2jp, slots 4-7: block 2jp+1)
    h16x8_t p1[2], p2[2];
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
      uint32_t w1[4], w2[4];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const f32x4_t& s = st[2 * jp + half];
        split2_pair(__builtin_ldexpf(s[0], P_EXP), __builtin_ldexpf(s[1], P_EXP), w1[half * 2], w2[half * 2]);
        split2_pair(__builtin_ldexpf(s[2], P_EXP), __builtin_ldexpf(s[3], P_EXP), w1[half * 2 + 1], w2[half * 2 + 1]);
      }
      p1[jp] = as_h8(w1[0], w1[1], w1[2], w1[3]);
      p2[jp] = as_h8(w2[0], w2[1], w2[2], w2[3]);
    }
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {
        const h16x8_t v1 = frag_vt(V1, d, jp, lane), v2 = frag_vt(V2, d, jp, lane);
        ot[d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(v1, p2[jp], ot[d], 0, 0, 0);
        ot[d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(v2, p1[jp], ot[d], 0, 0, 0);
        ot[d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(v1, p1[jp], ot[d], 0, 0, 0);
      }
  }
  if (qrow < p.T) {
    const float inv = o_unscale / l_i;
    float* orow = p.out + ((int64_t)b * p.T + qrow) * p.ldo + h * p.hd;
#pragma unroll
    for (int d = 0; d < NDB; ++d) {
      const int col = d * 16 + g * 4;
      if (col < p.hd)                                        // hd % 4 == 0: a quad is inside or outside as a whole
        *reinterpret_cast<float4*>(orow + col) = make_float4(ot[d][0] * inv, ot[d][1] * inv, ot[d][2] * inv, ot[d][3] * inv);
    }
  }
}

}  // namespace

extern "C" int ug_siglip_attn_f32(const float* qkv, int64_t ld, const float* qkv_amax, float* out, int64_t ldo, int64_t B,
                                  int64_t T, int H, int head_dim, float scale, hipStream_t st) {
  UG_REQUIRE(qkv && out && B > 0 && T > 0 && H > 0, "ug_siglip_attn_f32: bad args");
  UG_REQUIRE(head_dim > 0 && head_dim <= DP && head_dim % 4 == 0,
             "ug_siglip_attn_f32: head_dim %d unsupported (multiple of 4, at most %d)", head_dim, DP);
  UG_REQUIRE(ld % 4 == 0 && ldo % 4 == 0 && ug_aligned16(qkv) && ug_aligned16(out) && ld >= 3LL * H * head_dim && ldo >= (int64_t)H * head_dim,
             "ug_siglip_attn_f32: qkv [B*T, >= 3*H*head_dim] / out [B*T, >= H*head_dim] must be 16-byte aligned with row strides %% 4 == 0");
  SigArgs a{};
  a.qkv = qkv; a.out = out; a.amax = qkv_amax; a.ld = ld; a.ldo = ldo;
  a.B = (int)B; a.T = (int)T; a.H = H; a.hd = head_dim; a.scale = scale;
  hipLaunchKernelGGL(siglip_attn_kernel, dim3((unsigned)((T + 63) / 64), (unsigned)H, (unsigned)B), dim3(256), 0, st, a);
  UG_CHECK_LAUNCH("ug_siglip_attn_f32");
  return UG_OK;
}
