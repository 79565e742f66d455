// MAGVITv2 tokenizer kernels for gfx950 (reference: models/multimodal_encoder/magvitv2.py:57-442,
// common_modules.py:19-360).  The tokenizer runs in fp32 in the reference (outside autocast) and its
// outputs are SIGN BITS of 13 channels, so everything here is exact-fp32: convolutions are implicit
// GEMMs on the f32-input matrix cores (v_mfma_f32_32x32x2_f32 == an fp32 fma chain, 157 TF peak),
// GroupNorm statistics accumulate in fp64.
//
// Layout: activations NHWC fp32 (channels contiguous = the implicit-GEMM contraction is contiguous),
// weights repacked once at load time to [tap][Cin][CoutPad].  The NCHW<->NHWC conversion happens
// only at the API boundary (image in, reconstructed image out).
#include "common.h"
#include "unigen_hip.h"

namespace {

constexpr int CBM = 128;   // output pixels per workgroup
constexpr int CBK = 16;    // contraction slice (input channels of one tap)
constexpr int LDS_PAD = 4;

struct ConvArgs {
  const float* x;      // [B, Hin, Win, Cin]   (or A[M, K] with lda for plain GEMM)
  const float* w;      // [taps][Cin][ldw]     (or B)
  const float* bias;   // [Cout] or null
  const float* res;    // [M, ldres] residual added in the epilogue, or null
  float* y;            // [M, ldy]
  int B, Hin, Win, Cin, Hout, Wout, Cout;
  int KH, KW, stride, pad_t, pad_l, ups;   // ups: nearest-2x upsample of x folded into the load
  int ldw, ldy, ldres;
  int M;                                   // B*Hout*Wout
  // plain (batched) GEMM mode: x = A[M,K] row stride lda, w = B ([K][N] ldb, or [N][K] if b_nk)
  int gemm, lda, b_nk;
  int64_t sa, sb, sc;                      // batch strides (elements)
  int inner;                               // two-level batches: z = outer * inner + i; 0 = one level
  int64_t sa2, sb2, sc2;                   // strides of the outer level
  float alpha;
  int act;                                 // 0 none, 1 GELU(tanh) applied before the residual add
};

// Operand fetches are BRANCH-FREE: an invalid element is loaded from a safe dummy address (the operand's
// base) and replaced by zero afterwards, so the compiler issues all of a k-tile's global loads back to back
// (with guarded loads it serialised them behind `s_waitcnt vmcnt(0)`s).
// MODE 0: convolution (implicit im2col gather);  1: plain GEMM, B as [K][N];  2: plain GEMM, B as [N][K].
template <bool VEC4, int MODE>
__device__ __forceinline__ float4 load_a(const ConvArgs& p, const float* xb, int m, bool mvalid, int ob, int oy, int ox,
                                         int dy, int dx, int c, bool& keep) {
  const float* a;
  bool ok = mvalid;
  if constexpr (MODE != 0) {
    a = xb + (int64_t)m * p.lda + c;
  } else {
    int iy = oy * p.stride + dy - p.pad_t, ix = ox * p.stride + dx - p.pad_l;
    const int He = p.ups ? p.Hin * 2 : p.Hin, We = p.ups ? p.Win * 2 : p.Win;
    ok = ok && iy >= 0 && iy < He && ix >= 0 && ix < We;
    if (p.ups) { iy >>= 1; ix >>= 1; }
    a = xb + (((int64_t)ob * p.Hin + iy) * p.Win + ix) * p.Cin + c;
  }
  float4 v;
  if constexpr (VEC4) {        // Cin % 4 == 0: the quad is valid or not as a whole
    ok = ok && (c < p.Cin);
    v = *reinterpret_cast<const float4*>(ok ? a : xb);
    keep = ok;                  // zeroing is deferred to the LDS write so the load stays in flight under the MFMAs
  } else {
    keep = true;
    const bool o0 = ok && c < p.Cin, o1 = ok && c + 1 < p.Cin, o2 = ok && c + 2 < p.Cin, o3 = ok && c + 3 < p.Cin;
    const float t0 = *(o0 ? a : xb), t1 = *(o1 ? a + 1 : xb), t2 = *(o2 ? a + 2 : xb), t3 = *(o3 ? a + 3 : xb);
    v = make_float4(o0 ? t0 : 0.f, o1 ? t1 : 0.f, o2 ? t2 : 0.f, o3 ? t3 : 0.f);
  }
  return v;
}

// BN = 128: waves 2x2, each 64x64 (2x2 MFMA tiles);  BN = 32: waves 4x1, each 32x32
template <int BN, bool VEC4, int MODE>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(ConvArgs p) {
  constexpr int WN = (BN == 128) ? 2 : 1;          // waves along n
  constexpr int WM = 4 / WN;                       // waves along m
  constexpr int TM = CBM / WM / 32;                // 32x32 tiles per wave along m
  constexpr int TN = BN / WN / 32;
  constexpr int LDA = CBM + LDS_PAD, LDB = BN + LDS_PAD;
  __shared__ __attribute__((aligned(16))) float As[2][CBK * LDA];
  __shared__ __attribute__((aligned(16))) float Bs[2][CBK * LDB];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int m0 = blockIdx.x * CBM, n0 = blockIdx.y * BN;
  const int bz = p.inner ? (int)blockIdx.z % p.inner : (int)blockIdx.z, bo = p.inner ? (int)blockIdx.z / p.inner : 0;
  const float* xb = p.x + bz * p.sa + bo * p.sa2;
  const float* wb = p.w + bz * p.sb + bo * p.sb2;
  float* yb = p.y + bz * p.sc + bo * p.sc2;
  const float* rb = p.res ? p.res + bz * p.sc + bo * p.sc2 : nullptr;

  // this thread's two A pixels (rows tid>>2 and 64 + tid>>2), channel quad tid&3
  const int kc = tid & 3;
  int am[2], ab[2], ay[2], ax[2]; bool av[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + (tid >> 2) + i * 64;
    am[i] = m; av[i] = m < p.M;
    if constexpr (MODE == 0) {
      const int hw = p.Hout * p.Wout;
      const int mm = av[i] ? m : 0;
      ab[i] = mm / hw; const int r = mm % hw; ay[i] = r / p.Wout; ax[i] = r % p.Wout;
    } else { ab[i] = ay[i] = ax[i] = 0; }
  }
  const int cin_tiles = (p.Cin + CBK - 1) / CBK;
  const int taps = (MODE == 0) ? p.KH * p.KW : 1;
  const int nkt = taps * cin_tiles;

  f32x16_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[2], rbv[2];
  bool ka[2], kb[2];
  // Convolution operands are walked incrementally (tap outer, channel slice inner): the im2col coordinates, the
  // bounds test and the pixel's base pointer are recomputed only when the tap changes (every cin_tiles k-tiles), which
  // takes ~85 address VALU instructions out of every k-tile (PMC: 3.7 VALU per MFMA before).
  int cur_tap = 0, cur_c0 = -CBK;
  const float* apix[2] = {xb, xb};
  bool aok[2] = {false, false};
  auto fetch = [&](int kt) {
    int tap, c0;
    if constexpr (MODE == 0) {
      cur_c0 += CBK;
      if (cur_c0 >= p.Cin) { cur_c0 = 0; ++cur_tap; }
      tap = cur_tap; c0 = cur_c0;
      if (c0 == 0) {
        const int dy = tap / p.KW, dx = tap % p.KW;
        const int He = p.ups ? p.Hin * 2 : p.Hin, We = p.ups ? p.Win * 2 : p.Win;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          int iy = ay[i] * p.stride + dy - p.pad_t, ix = ax[i] * p.stride + dx - p.pad_l;
          aok[i] = av[i] && iy >= 0 && iy < He && ix >= 0 && ix < We;
          if (p.ups) { iy >>= 1; ix >>= 1; }
          apix[i] = aok[i] ? xb + (((int64_t)ab[i] * p.Hin + iy) * p.Win + ix) * p.Cin : xb;
        }
      }
    } else {
      tap = 0; c0 = kt * CBK;
    }
    if constexpr (MODE == 0 && VEC4) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int c = c0 + kc * 4;
        const bool ok = aok[i] && c < p.Cin;
        ra[i] = *reinterpret_cast<const float4*>(ok ? apix[i] + c : xb);
        ka[i] = ok;               // zeroing is deferred to the LDS write so the load stays in flight under the MFMAs
      }
    } else {
      const int dy = (MODE == 0) ? tap / p.KW : 0, dx = (MODE == 0) ? tap % p.KW : 0;
#pragma unroll
      for (int i = 0; i < 2; ++i) ra[i] = load_a<VEC4, MODE>(p, xb, am[i], av[i], ab[i], ay[i], ax[i], dy, dx, c0 + kc * 4, ka[i]);
    }
    if constexpr (MODE != 2) {      // B as [K][ldw]: float4 along n (rows are padded to the N tile by contract)
      constexpr int NB = (BN == 128) ? 2 : 1;
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int k = (BN == 128) ? (tid >> 5) + i * 8 : (tid >> 3);
        const int n4 = (BN == 128) ? (tid & 31) : (tid & 7);
        const bool ok = (c0 + k < p.Cin) && (BN == 128 || tid < 128);
        const float* src = wb + ((int64_t)tap * p.Cin + c0 + k) * p.ldw + n0 + n4 * 4;
        rbv[i] = *reinterpret_cast<const float4*>(ok ? src : wb);
        kb[i] = ok;
      }
    } else {                        // B as [N][K]: thread reads 4 consecutive k of one n (same pattern as A)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int nl = (tid >> 2) + i * 64;
        const int n = n0 + nl, c = c0 + kc * 4;
        const bool ok = n < p.Cout && nl < BN;
        const float* b = wb + (int64_t)n * p.ldw + c;
        if (VEC4 && (p.ldw & 3) == 0) {
          const bool o = ok && c < p.Cin;       // K % 4 == 0 whenever the vector path is taken for B
          rbv[i] = *reinterpret_cast<const float4*>(o ? b : wb);
          kb[i] = o;
        } else {
          kb[i] = true;
          const bool o0 = ok && c < p.Cin, o1 = ok && c + 1 < p.Cin, o2 = ok && c + 2 < p.Cin, o3 = ok && c + 3 < p.Cin;
          const float t0 = *(o0 ? b : wb), t1 = *(o1 ? b + 1 : wb), t2 = *(o2 ? b + 2 : wb), t3 = *(o3 ? b + 3 : wb);
          rbv[i] = make_float4(o0 ? t0 : 0.f, o1 ? t1 : 0.f, o2 ? t2 : 0.f, o3 ? t3 : 0.f);
        }
      }
    }
  };
  auto stash = [&](int buf) {
    float* a = As[buf]; float* b = Bs[buf];
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (!ka[i]) ra[i] = z4;
      if (i < ((MODE == 2 || BN == 128) ? 2 : 1) && !kb[i]) rbv[i] = z4;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ml = (tid >> 2) + i * 64;
      a[(kc * 4 + 0) * LDA + ml] = ra[i].x; a[(kc * 4 + 1) * LDA + ml] = ra[i].y;
      a[(kc * 4 + 2) * LDA + ml] = ra[i].z; a[(kc * 4 + 3) * LDA + ml] = ra[i].w;
    }
    if constexpr (MODE == 2) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int nl = (tid >> 2) + i * 64;
        if (nl < BN) {
          b[(kc * 4 + 0) * LDB + nl] = rbv[i].x; b[(kc * 4 + 1) * LDB + nl] = rbv[i].y;
          b[(kc * 4 + 2) * LDB + nl] = rbv[i].z; b[(kc * 4 + 3) * LDB + nl] = rbv[i].w;
        }
      }
    } else if constexpr (BN == 128) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int k = (tid >> 5) + i * 8, n4 = tid & 31;
        *reinterpret_cast<float4*>(b + k * LDB + n4 * 4) = rbv[i];
      }
    } else {
      if (tid < 128) { const int k = tid >> 3, n4 = tid & 7; *reinterpret_cast<float4*>(b + k * LDB + n4 * 4) = rbv[0]; }
    }
  };

  fetch(0);
  stash(0);
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nkt) fetch(kt + 1);             // global loads in flight under the MFMAs below
    const float* a = As[cur]; const float* b = Bs[cur];
#pragma unroll
    for (int kk = 0; kk < CBK / 2; ++kk) {
      const int krow = kk * 2 + (lane >> 5);
      float fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = a[krow * LDA + wm * (CBM / WM) + i * 32 + (lane & 31)];
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = b[krow * LDB + wn * (BN / WN) + j * 32 + (lane & 31)];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nkt) stash(cur ^ 1);            // other buffer: last read before the previous barrier
    __syncthreads();
  }

  // epilogue: C layout of the 32x32 tile: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * (BN / WN) + j * 32 + (lane & 31);
      if (n >= p.Cout) continue;
      const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * (CBM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m >= p.M) continue;
        float v = acc[i][j][r] * p.alpha + bv;
        if (p.act == 1) {   // gelu_pytorch_tanh: 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))
          const float u = 0.7978845608028654f * (v + 0.044715f * v * v * v);
          v = 0.5f * v * (1.f + tanhf(u));
        }
        if (rb) v += rb[(int64_t)m * p.ldres + n];
        yb[(int64_t)m * p.ldy + n] = v;
      }
    }
}

// ------------------------------------------------------------------ GroupNorm (32 groups) + swish
// stats[b][g] = {sum, sumsq} in fp64; x NHWC [B, HW, C]
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ x, double* __restrict__ stats, int HW,
                                                       int C, int G, int pix_per_block) {
  __shared__ double s1[32], s2[32];
  const int b = blockIdx.y;
  const int c4n = C >> 2;
  const int cpg = C / G;
  if (threadIdx.x < 32) { s1[threadIdx.x] = 0.0; s2[threadIdx.x] = 0.0; }
  __syncthreads();
  const int p0 = blockIdx.x * pix_per_block, p1 = min(HW, p0 + pix_per_block);
  // thread owns channel quad (tid % c4n) and walks pixels with stride 256 / c4n  (c4n divides 256 for C<=1024 pow2)
  const int cq = threadIdx.x % c4n, prow = threadIdx.x / c4n, pstep = 256 / c4n;
  double a1 = 0.0, a2 = 0.0;
  if (pstep > 0) {
    for (int px = p0 + prow; px < p1; px += pstep) {
      const float4 v = *reinterpret_cast<const float4*>(x + ((int64_t)b * HW + px) * C + cq * 4);
      a1 += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
      a2 += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    }
  }
  const int grp = (cq * 4) / cpg;     // cpg >= 4 so a quad never straddles groups
  atomicAdd(&s1[grp], a1);
  atomicAdd(&s2[grp], a2);
  __syncthreads();
  if (threadIdx.x < G) {
    atomicAdd(stats + ((int64_t)b * G + threadIdx.x) * 2 + 0, s1[threadIdx.x]);
    atomicAdd(stats + ((int64_t)b * G + threadIdx.x) * 2 + 1, s2[threadIdx.x]);
  }
}

__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x, const double* __restrict__ stats,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float* __restrict__ y, int64_t total4, int HW, int C, int G,
                                                       float eps, int swish) {
  const int c4n = C >> 2, cpg = C / G;
  const double n = (double)HW * cpg;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total4; idx += (int64_t)gridDim.x * blockDim.x) {
    const int cq = (int)(idx % c4n);
    const int64_t pix = idx / c4n;
    const int b = (int)(pix / HW);
    const int g = (cq * 4) / cpg;
    const double s1 = stats[((int64_t)b * G + g) * 2], s2 = stats[((int64_t)b * G + g) * 2 + 1];
    const double mean = s1 / n;
    double var = s2 / n - mean * mean; if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float mu = (float)mean;
    const float4 v = reinterpret_cast<const float4*>(x)[idx];
    const float4 ga = reinterpret_cast<const float4*>(gamma)[cq], be = reinterpret_cast<const float4*>(beta)[cq];
    float o[4] = {(v.x - mu) * rstd * ga.x + be.x, (v.y - mu) * rstd * ga.y + be.y,
                  (v.z - mu) * rstd * ga.z + be.z, (v.w - mu) * rstd * ga.w + be.w};
    if (swish) {
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = o[k] / (1.f + expf(-o[k]));   // x * sigmoid(x)
    }
    reinterpret_cast<float4*>(y)[idx] = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// (mean, rstd) per (image, group) in fp32, rounded exactly as gn_apply_kernel rounds them: lets a consumer apply the
// normalisation on its own load path
__global__ __launch_bounds__(256) void gn_finalize_kernel(const double* __restrict__ stats, float2* __restrict__ mu_rstd,
                                                          int n_stats, double n, float eps) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_stats) return;
  const double mean = stats[2 * i] / n;
  double var = stats[2 * i + 1] / n - mean * mean; if (var < 0.0) var = 0.0;
  mu_rstd[i] = make_float2((float)mean, (float)(1.0 / sqrt(var + (double)eps)));
}

// ------------------------------------------------------------------ row softmax (AttnBlock), fp32
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ x, int rows, int cols, int64_t ld, float scale) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  float* r = x + (int64_t)row * ld;
  float m = -INFINITY;
  for (int c = lane; c < cols; c += 64) m = fmaxf(m, r[c] * scale);
  m = wave_max(m);
  float s = 0.f;
  for (int c = lane; c < cols; c += 64) { const float e = expf(r[c] * scale - m); r[c] = e; s += e; }
  s = wave_sum(s);
  const float inv = 1.f / s;
  for (int c = lane; c < cols; c += 64) r[c] *= inv;
  // columns up to the next multiple of four (when the row has them) read as zero, so a consumer may contract over the
  // padded width with 16-byte loads
  const int padded = min((int)ld, (cols + 3) & ~3);
  if (cols + lane < padded) r[cols + lane] = 0.f;
}

// ------------------------------------------------------------------ LayerNorm (SigLIP, eps 1e-6), fp32, one wave per row
__global__ __launch_bounds__(256) void layernorm_f32_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                            const float* __restrict__ b, float* __restrict__ y, int rows,
                                                            int cols, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* r = x + (int64_t)row * cols;
  float s = 0.f;
  for (int c = lane; c < cols; c += 64) s += r[c];
  const float mean = wave_sum(s) / (float)cols;
  float v = 0.f;
  for (int c = lane; c < cols; c += 64) { const float d = r[c] - mean; v += d * d; }
  const float rstd = rsqrtf(wave_sum(v) / (float)cols + eps);
  float* o = y + (int64_t)row * cols;
  for (int c = lane; c < cols; c += 64) o[c] = (r[c] - mean) * rstd * g[c] + b[c];
}

// ------------------------------------------------------------------ layout conversion at the API boundary
// NCHW [B,C,H,W] -> NHWC [B,H,W,Cp] (channels C..Cp-1 zero)
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out, int B,
                                                           int C, int HW, int Cp) {
  const int64_t total = (int64_t)B * HW * Cp;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % Cp);
    const int64_t pix = idx / Cp;
    const int b = (int)(pix / HW); const int hw = (int)(pix % HW);
    out[idx] = (c < C) ? in[((int64_t)b * C + c) * HW + hw] : 0.f;
  }
}
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ in, float* __restrict__ out, int B,
                                                           int C, int HW, int Cp) {
  const int64_t total = (int64_t)B * C * HW;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int hw = (int)(idx % HW);
    const int c = (int)((idx / HW) % C);
    const int b = (int)(idx / ((int64_t)HW * C));
    out[idx] = in[((int64_t)b * HW + hw) * Cp + c];
  }
}

// ------------------------------------------------------------------ lookup-free quantiser
// index = sum_c 2^(nbits-1-c) * [z_c > 0]   (LFQuantizer.get_indices, magvitv2.py:210-215; channel 0 = MSB)
__global__ __launch_bounds__(256) void lfq_pack_kernel(const float* __restrict__ z, int64_t ldz, int64_t* __restrict__ idx,
                                                       int64_t n, int nbits) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  int64_t code = 0;
  for (int c = 0; c < nbits; ++c) code |= (int64_t)(z[i * ldz + c] > 0.f) << (nbits - 1 - c);
  idx[i] = code;
}
// codes -> +-1 entries  (LFQuantizer.get_codebook_entry, magvitv2.py:217-230), NHWC [n, nbits]
__global__ __launch_bounds__(256) void lfq_unpack_kernel(const int64_t* __restrict__ idx, float* __restrict__ z, int64_t n,
                                                         int nbits, int* __restrict__ err) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  int64_t code = idx[i];
  if (code < 0 || code >= (1LL << nbits)) { if (err) atomicOr(err, 4); code = 0; }
  for (int c = 0; c < nbits; ++c) z[i * nbits + c] = ((code >> (nbits - 1 - c)) & 1) ? 1.f : -1.f;
}

template <int MODE>
void launch_conv_mode(const ConvArgs& a, int nb, bool vec4, hipStream_t st) {
  const bool wide = a.Cout > 32;
  dim3 block(256);
  if (wide) {
    dim3 grid((a.M + CBM - 1) / CBM, (a.Cout + 127) / 128, nb);
    if (vec4) hipLaunchKernelGGL((conv_igemm_kernel<128, true, MODE>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((conv_igemm_kernel<128, false, MODE>), grid, block, 0, st, a);
  } else {
    dim3 grid((a.M + CBM - 1) / CBM, 1, nb);
    if (vec4) hipLaunchKernelGGL((conv_igemm_kernel<32, true, MODE>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((conv_igemm_kernel<32, false, MODE>), grid, block, 0, st, a);
  }
}

// pixels per statistics workgroup: ~2048 workgroups per launch (a 16-image 32x32x512 tensor used to run on 16 of them)
int gn_pixels_per_block(int64_t B, int64_t HW) {
  int64_t ppb = (B * HW + 2047) / 2048;
  ppb = (ppb + 63) / 64 * 64;
  return (int)(ppb < 64 ? 64 : ppb > 1024 ? 1024 : ppb);
}

int launch_conv(const ConvArgs& a, int nb, hipStream_t st) {
  // vector path: every 4-element A fetch is 16-byte aligned and never straddles the contraction extent
  const bool vec4 = a.gemm ? (a.lda % 4 == 0 && a.Cin % 4 == 0 && ug_aligned16(a.x) && a.sa % 4 == 0 && a.sa2 % 4 == 0 &&
                              (!a.b_nk || (a.ldw % 4 == 0 && ug_aligned16(a.w) && a.sb % 4 == 0 && a.sb2 % 4 == 0)))
                           : (a.Cin % 4 == 0 && ug_aligned16(a.x));
  if (!a.gemm) launch_conv_mode<0>(a, nb, vec4, st);
  else if (!a.b_nk) launch_conv_mode<1>(a, nb, vec4, st);
  else launch_conv_mode<2>(a, nb, vec4, st);
  return UG_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------ C ABI
extern "C" int ug_conv2d_f32(const float* x, const float* w_packed, const float* bias, const float* residual, float* y,
                             int64_t B, int Hin, int Win, int Cin, int Cout, int cout_pad, int ksize, int stride,
                             int pad_top, int pad_left, int Hout, int Wout, int upsample2x, hipStream_t st) {
  UG_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && ksize >= 1 && ksize <= 16, "ug_conv2d_f32: bad shape");
  UG_REQUIRE(cout_pad % 4 == 0 && cout_pad >= ((Cout > 32) ? (Cout + 127) / 128 * 128 : 32),
             "ug_conv2d_f32: packed weights must be padded to the N tile (cout_pad=%d for Cout=%d)", cout_pad, Cout);
  UG_REQUIRE(ug_aligned16(w_packed) && x && y, "ug_conv2d_f32: pointers");
  ConvArgs a{};
  a.x = x; a.w = w_packed; a.bias = bias; a.res = residual; a.y = y;
  a.B = (int)B; a.Hin = Hin; a.Win = Win; a.Cin = Cin; a.Hout = Hout; a.Wout = Wout; a.Cout = Cout;
  a.KH = ksize; a.KW = ksize; a.stride = stride; a.pad_t = pad_top; a.pad_l = pad_left; a.ups = upsample2x;
  a.ldw = cout_pad; a.ldy = Cout; a.ldres = Cout;
  const int64_t M = B * Hout * Wout;
  UG_REQUIRE(M < (1LL << 31), "ug_conv2d_f32: too many output pixels");
  a.M = (int)M; a.gemm = 0; a.alpha = 1.f;
  launch_conv(a, 1, st);
  UG_CHECK_LAUNCH("ug_conv2d_f32");
  return UG_OK;
}

extern "C" int ug_gemm_f32(const float* A, int64_t lda, int64_t stride_a, const float* Bm, int64_t ldb, int64_t stride_b,
                           int b_is_nk, float* C, int64_t ldc, int64_t stride_c, int64_t M, int64_t N, int64_t K,
                           int64_t batch, float alpha, hipStream_t st) {
  UG_REQUIRE(M > 0 && N > 0 && K > 0 && batch > 0, "ug_gemm_f32: empty");
  UG_REQUIRE(b_is_nk || (ldb % 4 == 0 && ug_aligned16(Bm) && stride_b % 4 == 0 && ldb >= ((N > 32) ? (N + 127) / 128 * 128 : 32)),
             "ug_gemm_f32: [K][N] operand needs ldb padded to the N tile and 16B alignment");
  ConvArgs a{};
  a.x = A; a.w = Bm; a.y = C; a.Cin = (int)K; a.Cout = (int)N; a.M = (int)M;
  a.gemm = 1; a.lda = (int)lda; a.ldw = (int)ldb; a.ldy = (int)ldc; a.b_nk = b_is_nk;
  a.sa = stride_a; a.sb = stride_b; a.sc = stride_c; a.alpha = alpha; a.KH = a.KW = 1;
  launch_conv(a, (int)batch, st);
  UG_CHECK_LAUNCH("ug_gemm_f32");
  return UG_OK;
}

extern "C" int ug_gemm_f32_nested(const float* A, int64_t lda, int64_t sa_in, int64_t sa_out, const float* Bm, int64_t ldb,
                                  int64_t sb_in, int64_t sb_out, int b_is_nk, float* C, int64_t ldc, int64_t sc_in,
                                  int64_t sc_out, int64_t M, int64_t N, int64_t K, int64_t batch_in, int64_t batch_out,
                                  float alpha, hipStream_t st) {
  UG_REQUIRE(M > 0 && N > 0 && K > 0 && batch_in > 0 && batch_out > 0 && batch_in * batch_out < 65536, "ug_gemm_f32_nested: bad sizes");
  UG_REQUIRE(b_is_nk || (ldb % 4 == 0 && ug_aligned16(Bm) && sb_in % 4 == 0 && sb_out % 4 == 0 &&
                         ldb >= ((N > 32) ? (N + 127) / 128 * 128 : 32)),
             "ug_gemm_f32_nested: [K][N] operand needs ldb padded to the N tile and 16B alignment");
  ConvArgs a{};
  a.x = A; a.w = Bm; a.y = C; a.Cin = (int)K; a.Cout = (int)N; a.M = (int)M;
  a.gemm = 1; a.lda = (int)lda; a.ldw = (int)ldb; a.ldy = (int)ldc; a.b_nk = b_is_nk;
  a.sa = sa_in; a.sb = sb_in; a.sc = sc_in; a.inner = (int)batch_in; a.sa2 = sa_out; a.sb2 = sb_out; a.sc2 = sc_out;
  a.alpha = alpha; a.KH = a.KW = 1;
  launch_conv(a, (int)(batch_in * batch_out), st);
  UG_CHECK_LAUNCH("ug_gemm_f32_nested");
  return UG_OK;
}

extern "C" int ug_linear_f32(const float* x, int64_t ldx, const float* W, int64_t ldw, const float* bias,
                             const float* residual, int64_t ldres, float* y, int64_t ldy, int64_t M, int64_t N, int64_t K,
                             int act, hipStream_t st) {
  UG_REQUIRE(M > 0 && N > 0 && K > 0 && (act == 0 || act == 1), "ug_linear_f32: bad args");
  UG_REQUIRE(M < (1LL << 31), "ug_linear_f32: too many rows");
  ConvArgs a{};
  a.x = x; a.w = W; a.bias = bias; a.res = residual; a.y = y; a.Cin = (int)K; a.Cout = (int)N; a.M = (int)M;
  a.gemm = 1; a.lda = (int)ldx; a.ldw = (int)ldw; a.ldy = (int)ldy; a.ldres = (int)ldres; a.b_nk = 1;     // W is [N][K] like nn.Linear
  a.alpha = 1.f; a.act = act; a.KH = a.KW = 1;
  launch_conv(a, 1, st);
  UG_CHECK_LAUNCH("ug_linear_f32");
  return UG_OK;
}

extern "C" int ug_layernorm_f32(const float* x, const float* gamma, const float* beta, float* y, int64_t rows, int64_t cols,
                                float eps, hipStream_t st) {
  UG_REQUIRE(rows > 0 && cols > 0, "ug_layernorm_f32: empty");
  hipLaunchKernelGGL(layernorm_f32_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, x, gamma, beta, y, (int)rows,
                     (int)cols, eps);
  UG_CHECK_LAUNCH("ug_layernorm_f32");
  return UG_OK;
}

extern "C" int ug_groupnorm_swish(const float* x, const float* gamma, const float* beta, float* y, double* stats_ws,
                                  int64_t B, int64_t HW, int C, int groups, float eps, int apply_swish, hipStream_t st) {
  UG_REQUIRE(B > 0 && HW > 0 && groups > 0 && groups <= 32 && C % groups == 0 && (C / groups) % 4 == 0,
             "ug_groupnorm_swish: unsupported C=%d groups=%d", C, groups);
  UG_REQUIRE(C / 4 <= 256 && 256 % (C / 4) == 0, "ug_groupnorm_swish: C=%d must be 4*2^k <= 1024", C);
  UG_HIP(hipMemsetAsync(stats_ws, 0, sizeof(double) * 2 * B * groups, st));
  const int ppb = gn_pixels_per_block(B, HW);
  dim3 grid((unsigned)((HW + ppb - 1) / ppb), (unsigned)B);
  hipLaunchKernelGGL(gn_stats_kernel, grid, dim3(256), 0, st, x, stats_ws, (int)HW, C, groups, ppb);
  UG_CHECK_LAUNCH("ug_groupnorm_swish(stats)");
  const int64_t total4 = B * HW * (C / 4);
  int64_t g = (total4 + 255) / 256; if (g > 256 * 16) g = 256 * 16;
  hipLaunchKernelGGL(gn_apply_kernel, dim3((unsigned)g), dim3(256), 0, st, x, stats_ws, gamma, beta, y, total4, (int)HW, C,
                     groups, eps, apply_swish);
  UG_CHECK_LAUNCH("ug_groupnorm_swish(apply)");
  return UG_OK;
}

extern "C" int ug_groupnorm_stats(const float* x, double* stats_ws, float* mu_rstd, int64_t B, int64_t HW, int C,
                                  int groups, float eps, hipStream_t st) {
  UG_REQUIRE(B > 0 && HW > 0 && groups > 0 && groups <= 32 && C % groups == 0 && (C / groups) % 4 == 0,
             "ug_groupnorm_stats: unsupported C=%d groups=%d", C, groups);
  UG_REQUIRE(C / 4 <= 256 && 256 % (C / 4) == 0, "ug_groupnorm_stats: C=%d must be 4*2^k <= 1024", C);
  UG_REQUIRE(x && stats_ws && mu_rstd && ((uintptr_t)mu_rstd & 7) == 0, "ug_groupnorm_stats: pointers");
  UG_HIP(hipMemsetAsync(stats_ws, 0, sizeof(double) * 2 * B * groups, st));
  const int ppb = gn_pixels_per_block(B, HW);
  dim3 grid((unsigned)((HW + ppb - 1) / ppb), (unsigned)B);
  hipLaunchKernelGGL(gn_stats_kernel, grid, dim3(256), 0, st, x, stats_ws, (int)HW, C, groups, ppb);
  UG_CHECK_LAUNCH("ug_groupnorm_stats(stats)");
  const int n_stats = (int)(B * groups);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3((unsigned)((n_stats + 255) / 256)), dim3(256), 0, st, stats_ws,
                     reinterpret_cast<float2*>(mu_rstd), n_stats, (double)HW * (C / groups), eps);
  UG_CHECK_LAUNCH("ug_groupnorm_stats(finalize)");
  return UG_OK;
}

extern "C" int ug_groupnorm_finalize(const double* stats, float* mu_rstd, int64_t B, int64_t HW, int C, int groups, float eps,
                                     hipStream_t st) {
  UG_REQUIRE(B > 0 && HW > 0 && groups > 0 && C % groups == 0 && stats && mu_rstd && ((uintptr_t)mu_rstd & 7) == 0,
             "ug_groupnorm_finalize: bad args (C=%d groups=%d)", C, groups);
  const int n_stats = (int)(B * groups);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3((unsigned)((n_stats + 255) / 256)), dim3(256), 0, st, stats,
                     reinterpret_cast<float2*>(mu_rstd), n_stats, (double)HW * (C / groups), eps);
  UG_CHECK_LAUNCH("ug_groupnorm_finalize");
  return UG_OK;
}

extern "C" int ug_softmax_rows_f32(float* x, int64_t rows, int64_t cols, int64_t ld, float scale, hipStream_t st) {
  UG_REQUIRE(rows > 0 && cols > 0 && ld >= cols, "ug_softmax_rows_f32: bad shape");
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, x, (int)rows, (int)cols, ld, scale);
  UG_CHECK_LAUNCH("ug_softmax_rows_f32");
  return UG_OK;
}

extern "C" int ug_nchw_to_nhwc(const float* in, float* out, int64_t B, int C, int64_t HW, int c_pad, hipStream_t st) {
  UG_REQUIRE(B > 0 && C > 0 && c_pad >= C, "ug_nchw_to_nhwc: bad args");
  const int64_t total = B * HW * c_pad; int64_t g = (total + 255) / 256; if (g > 4096) g = 4096;
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((unsigned)g), dim3(256), 0, st, in, out, (int)B, C, (int)HW, c_pad);
  UG_CHECK_LAUNCH("ug_nchw_to_nhwc");
  return UG_OK;
}
extern "C" int ug_nhwc_to_nchw(const float* in, float* out, int64_t B, int C, int64_t HW, int c_pad, hipStream_t st) {
  UG_REQUIRE(B > 0 && C > 0 && c_pad >= C, "ug_nhwc_to_nchw: bad args");
  const int64_t total = B * HW * C; int64_t g = (total + 255) / 256; if (g > 4096) g = 4096;
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((unsigned)g), dim3(256), 0, st, in, out, (int)B, C, (int)HW, c_pad);
  UG_CHECK_LAUNCH("ug_nhwc_to_nchw");
  return UG_OK;
}

extern "C" int ug_lfq_pack(const float* z, int64_t ldz, int64_t* idx, int64_t n, int nbits, hipStream_t st) {
  UG_REQUIRE(n > 0 && nbits > 0 && nbits < 63, "ug_lfq_pack: bad args");
  hipLaunchKernelGGL(lfq_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, z, ldz, idx, n, nbits);
  UG_CHECK_LAUNCH("ug_lfq_pack");
  return UG_OK;
}
extern "C" int ug_lfq_unpack(const int64_t* idx, float* z, int64_t n, int nbits, int* err_flag, hipStream_t st) {
  UG_REQUIRE(n > 0 && nbits > 0 && nbits < 63, "ug_lfq_unpack: bad args");
  hipLaunchKernelGGL(lfq_unpack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, idx, z, n, nbits, err_flag);
  UG_CHECK_LAUNCH("ug_lfq_unpack");
  return UG_OK;
}
