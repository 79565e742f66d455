// MAGVITv2 tokenizer convolutions (and SigLIP projections) at fp32 accuracy on the 16-bit matrix cores (reference:
// common_modules.py:19-360, magvitv2.py:57-200 -- every nn.Conv2d of VQGANEncoder / VQGANDecoder runs in fp32 there).
//
// gfx950's f16 MFMA runs 16x the rate of its fp32 MFMA, so an fp32 product is rebuilt from fp16 pieces.  Each operand
// tensor is first scaled by a power of two that puts its largest magnitude into [2^14, 2^15) (exact; the exponent comes
// from an upper bound of max|x| supplied by the caller / computed at weight-pack time), then every value is split into
// two fp16 terms a = a1 + a2 (a1 = RNE(a), a2 = RNE(a - a1): 11 + 11 significand bits plus the sign of a2 carry 23 of
// fp32's 24 bits; a2 of a value far below the tensor's maximum goes subnormal, an absolute error of 2^-39 of that
// maximum), and a.b is summed from the three partial products a1b1 + a1b2 + a2b1, each exact in the fp32 accumulator;
// the dropped a2b2 is <= 2^-22 |a.b|.  Measured against fp64 the result is as close as a plain fp32 accumulation
// (relative error 3.5e-7 at K = 1152, the same as an fp32 matmul; the reference's own GPU path uses TF32 operands,
// 7.7e-4).  Three f16 MFMAs cost 3/16 of one fp32 MFMA: the bound moves from 157 TF/s to ~830 TF/s fp32-equivalent.
// (Round 1 used three bf16 terms and six products: exact operand split, twice the MFMA work, error 1.7e-7.)
//
// Implicit GEMM  y[m, n] = sum_{tap, c} x[pix(m, tap), c] * w[tap][c][n]  with the WEIGHTS as the MFMA A operand
// (rows = cout) and the PIXELS as the B operand, so that each lane of a 16x16 result block holds four consecutive
// output channels of one pixel (one float4 store, NHWC).  Weights are scaled and split once at pack time
// (ug_conv_split_weights) into the exact LDS image of each (tap, 32-channel slab, 128-cout block) tile; activations
// are scaled and split on the way from registers to LDS; the epilogue multiplies the accumulator by 2^-(ex + ew).
#include "common.h"
#include "split_f16.h"
#include "unigen_hip.h"

namespace {

constexpr int SBM = 128;                 // output pixels per workgroup
constexpr int SBN = 128;                 // output channels per workgroup
constexpr int SBK = 32;                  // contraction slab = one 16x16x32 MFMA step
constexpr int PLANE = SBN * SBK;         // 16-bit elements of one plane of one tile (128 rows x 32 k)
constexpr int NPL = 2;                   // planes: x1 = RNE_f16(x), x2 = RNE_f16(x - x1)
constexpr int TILE = NPL * PLANE;
struct SplitArgs {
  const float* x;          // [B, Hin, Win, Cin] fp32 NHWC
  const bf16_t* w;         // split tiles (raw fp16 bits), see ug_conv_split_weights
  const float* w_amax;     // max|w| recorded behind the tiles at pack time
  const float* x_amax;     // upper bound of max|x| (device) or null
  const float* bias;       // [Cout] or null
  const float* res;        // [M, Cout] or null
  float* y;                // [M, Cout]
  int B, Hin, Win, Cin, Hout, Wout, Cout;
  int KH, KW, stride, pad_t, pad_l, ups;
  int nblks;               // cout_pad / 128
  int M;
  int kslabs;              // ceil(Cin / 32): a ragged last slab reads zeros (weights are packed zero-padded)
  int64_t ldx, ldy, ldres; // row strides of x (per input pixel), y and residual (per output pixel), elements
  int act;                 // 1: gelu_pytorch_tanh before the residual add
  double* stats_out;       // [B][Cout / out_cpg][2] sums of y for the GroupNorm that reads it (see QuadStats), or null;
  int out_cpg;             // needs Hout * Wout % 128 == 0 (a pixel tile inside one image)
  float* amax_out;         // max |y| (with stats_out)
};

// LDS image of one plane: row r (64 B), 16-byte chunk c (8 k) stored at chunk c ^ swz_q(r).  A ds_read_b128 is
// served in four groups of 16 lanes that are NOT consecutive -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and their upper
// halves (MI355X_MICROARCH.md, LDS table) -- i.e. rows 0-3 and 12-15 of chunk g together with rows 4-11 of chunk g^1; the
// row-quad map has to put each group's 16 pieces on 16 distinct 16-byte slots of the 256-byte bank row (the identity map of
// round 1 put rows 0-3 / chunk g and rows 4-7 / chunk g^1 on the same slots: SQ_LDS_BANK_CONFLICT 1.7e8 on the patch kernel;
// round 2's (0, 3, 2, 1), the 256x256 GEMM's map, does so only for reads that start at a multiple of 16 rows).
// Round 3: the quad map is (0, 2, 0, 2), i.e. chunk ^= ((r >> 2) & 1) << 1, not (0, 3, 2, 1).  The old map is conflict-free only for
// fragment reads whose 16 rows start at a multiple of 16 (weight tiles, im2col tiles); the LDS-resident patch reads 16 pixel rows
// starting at (y + dy) * 18 + x + dx, and for 224 of the 256 (start mod 64, lane group) combinations two of a service group's 16
// pieces met on one slot (SQ_LDS_BANK_CONFLICT 7.9e7 on the 16-row kernel, ~6-8 % of its cycles).  A service group reads rows
// {i, i + 12} of chunk g and {i + 4, i + 8} of chunk g ^ 1 in every row-residue class; with s(q) the map of quad q the four slots
// s(q), s(q + 1) ^ 1, s(q + 2) ^ 1, s(q + 3) are distinct for EVERY q exactly for the period-2 maps (0, 2), (0, 3), (1, 2), ... --
// enumerated over all 4^4 maps.
__device__ __forceinline__ int swz_q(int r) { return ((r >> 2) & 1) << 1; }
__device__ __forceinline__ int swz(int r, int chunk) { return r * SBK + ((chunk ^ swz_q(r)) << 3); }

// four fp32 values of one pixel (channel quad `q` of LDS row `row`), scaled by 2^ex -> the two split planes
__device__ __forceinline__ void store_split_quad(bf16_t* planes, int plane_stride, int row, int q, f32x4_t v, int ex) {
  uint32_t a1, a2, b1, b2;
  split2_pair(__builtin_ldexpf(v[0], ex), __builtin_ldexpf(v[1], ex), a1, a2);
  split2_pair(__builtin_ldexpf(v[2], ex), __builtin_ldexpf(v[3], ex), b1, b2);
  const int off = swz(row, q >> 1) + (q & 1) * 4;
  *reinterpret_cast<uint2*>(planes + off) = make_uint2(a1, b1);
  *reinterpret_cast<uint2*>(planes + plane_stride + off) = make_uint2(a2, b2);
}

// result block epilogue: four consecutive output channels of one pixel.  y = act(acc * 2^-(ex+ew) + bias) + residual
__device__ __forceinline__ f32x4_t store_out_quad(f32x4_t a, float unscale, const float* bias, const float* res, float* y, int act) {
  float v[4] = {a[0] * unscale, a[1] * unscale, a[2] * unscale, a[3] * unscale};
  if (bias) {
    const float4 b = *reinterpret_cast<const float4*>(bias);
    v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
  }
  if (act == 1) {            // gelu_pytorch_tanh: 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3))), as ug_linear_f32
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float u = 0.7978845608028654f * (v[k] + 0.044715f * v[k] * v[k] * v[k]);
      v[k] = 0.5f * v[k] * (1.f + tanhf(u));
    }
  }
  if (res) {
    const float4 r = *reinterpret_cast<const float4*>(res);
    v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
  }
  *reinterpret_cast<float4*>(y) = make_float4(v[0], v[1], v[2], v[3]);
  return f32x4_t{v[0], v[1], v[2], v[3]};
}

// Statistics of the NEXT GroupNorm gathered while the output is stored (the consumer of almost every 3x3 convolution of a
// ResnetBlock is a GroupNorm: common_modules.py:308-335): fp64 sum and sum of squares of the stored fp32 values, exactly the
// arithmetic of gn_stats_kernel (conv_f32.hip) -- a lane's channel quad lies inside one group (channels per group >= 4) --
// reduced over the 16 pixel lanes, then over the workgroup through 2 x 32 LDS slots (one per quad of its 128 channels), then
// ONE fp64 atomic per quad and statistic into stats[b][group][2].  Saves the separate pass over the tensor (537 MB at 256^2).
struct QuadStats {
  double s1, s2;
  __device__ __forceinline__ void add(f32x4_t v) {
    s1 += ((double)v[0] + (double)v[1]) + ((double)v[2] + (double)v[3]);
    s2 += ((double)v[0] * v[0] + (double)v[1] * v[1]) + ((double)v[2] * v[2] + (double)v[3] * v[3]);
  }
};
__device__ __forceinline__ float quad_absmax(f32x4_t v) {
  return fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
}
__device__ __forceinline__ double shfl_xor_f64(double v, int m) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __shfl_xor(lo, m, 64); hi = __shfl_xor(hi, m, 64);
  return __hiloint2double(hi, lo);
}
// qs[i]: this lane's sums for output block i (quad index within the workgroup's channel range: qbase + i * 4 + g)
// mx: max |y| over this lane's stored values -> *amax_out (order-preserving unsigned max of the float bits, as amax_kernel):
// the scale bound a following split convolution WITHOUT a GroupNorm on its load path (Downsample, nin_shortcut) needs of its
// input, without the pass over the tensor.
template <int NI>
__device__ __forceinline__ void flush_quad_stats(QuadStats (&qs)[NI], float mx, double* sred, int qbase, int g, int l16, int tid,
                                                 int n0, int Cout, int out_cpg, double* stats_b, float* amax_out) {
  if (tid < 64) sred[tid] = 0.0;
  unsigned int* mred = reinterpret_cast<unsigned int*>(sred + 64);
  if (tid == 0) *mred = 0u;
  __syncthreads();
  mx = wave_max(mx);
  if ((tid & 63) == 0) atomicMax(mred, __float_as_uint(mx));
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    double a = qs[i].s1, b = qs[i].s2;
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) { a += shfl_xor_f64(a, m); b += shfl_xor_f64(b, m); }
    if (l16 == 0) {
      atomicAdd(&sred[2 * (qbase + i * 4 + g)], a);
      atomicAdd(&sred[2 * (qbase + i * 4 + g) + 1], b);
    }
  }
  __syncthreads();
  if (tid < 64) {
    const int n = n0 + (tid >> 1) * 4;                    // first channel of quad tid / 2
    if (n < Cout) atomicAdd(stats_b + (n / out_cpg) * 2 + (tid & 1), sred[tid]);
  }
  if (tid == 64 && *mred) atomicMax(reinterpret_cast<unsigned int*>(amax_out), *mred);
}

// GroupNorm (+ swish) of a channel quad: gn_apply_kernel's arithmetic (conv_f32.hip) with the sigmoid on the hardware
// exp2 / reciprocal (~3e-7 relative, far below the summation noise of the convolution that consumes it; the accurate
// expf + IEEE division cost a tenth of the fused kernels' time)
__device__ __forceinline__ f32x4_t gn_swish_quad(f32x4_t v, float mu, float rstd, f32x4_t ga, f32x4_t be, int swish) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float o = (v[k] - mu) * rstd * ga[k] + be[k];
    if (swish) o = o * __builtin_amdgcn_rcpf(1.f + __expf(-o));
    v[k] = o;
  }
  return v;
}

// One 32-deep contraction slab of a wave's 64 x 64 result (4 x 4 MFMA blocks): three partial products per block, the
// small ones first.  wl: this lane's row of the weight plane-0 image (blocks 16 rows apart); xj[j]: this lane's row of
// pixel block j in plane 0; planes are XPLANE (pixels) / PLANE (weights) elements apart.
template <int XPLANE, int NI = 4>
__device__ __forceinline__ void mma_slab(const bf16_t* wl, const bf16_t* const (&xj)[4], f32x4_t (&acc)[NI][4]) {
  h16x8_t w1[NI], w2[NI], x1[4], x2[4];
#pragma unroll
  for (int i = 0; i < NI; ++i) w1[i] = *reinterpret_cast<const h16x8_t*>(wl + i * 16 * SBK);
#pragma unroll
  for (int j = 0; j < 4; ++j) x2[j] = *reinterpret_cast<const h16x8_t*>(xj[j] + XPLANE);
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[i], x2[j], acc[i][j], 0, 0, 0);
#pragma unroll
  for (int i = 0; i < NI; ++i) w2[i] = *reinterpret_cast<const h16x8_t*>(wl + PLANE + i * 16 * SBK);
#pragma unroll
  for (int j = 0; j < 4; ++j) x1[j] = *reinterpret_cast<const h16x8_t*>(xj[j]);
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2[i], x1[j], acc[i][j], 0, 0, 0);
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[i], x1[j], acc[i][j], 0, 0, 0);
    }
}

// NI: 16-channel output blocks per wave (4: 128 output channels per workgroup; 2: 64, for outputs whose 128-wide tiling
// leaves CUs idle -- SigLIP's 1152-wide projections at four images, the tokenizer's 1x1 convs at 16 x 16)
template <int NI>
__global__ __launch_bounds__(256, 2) void conv_split_kernel(SplitArgs p) {
  __shared__ __attribute__((aligned(16))) bf16_t Ws[TILE];
  __shared__ __attribute__((aligned(16))) bf16_t Xs[TILE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave & 1, wm = wave >> 1;
  const int g = lane >> 4, l16 = lane & 15;
  const int ex = scale_exp(p.x_amax);
  const float unscale = __builtin_ldexpf(1.f, -(ex + scale_exp(p.w_amax)));
  // consecutive workgroup ids land on different XCDs (8 L2s): give each XCD a contiguous band of pixel tiles so the
  // rows two vertically adjacent tiles share are fetched into one L2, not three
  const int per_xcd = gridDim.x >> 3;
  const int mt = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  const int m0 = mt * SBM;
  if (m0 >= p.M) return;
  const int nblk = NI == 4 ? blockIdx.y : blockIdx.y >> 1, nhalf = NI == 4 ? 0 : blockIdx.y & 1;

  // activation fetch: thread owns channel quad q of pixel rows (tid >> 3) + 32 i
  const int q = tid & 7;
  int ab[4], ay[4], ax[4];
  bool av[4];
  const int hw = p.Hout * p.Wout;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + (tid >> 3) + i * 32;
    av[i] = m < p.M;
    const int mm = av[i] ? m : 0;
    ab[i] = mm / hw;
    const int r = mm % hw;
    ay[i] = r / p.Wout;
    ax[i] = r % p.Wout;
  }
  const int nkt = p.KH * p.KW * p.kslabs;

  f32x4_t acc[NI][4];                    // [n block][m block]
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  float4 rx[4];
  bool kx[4];
  constexpr int WPIECES = NI == 4 ? 4 : 2;                  // NI == 2: this workgroup's 64 rows of each weight plane
  constexpr int WSTEP = NI == 4 ? 256 : PLANE / 8;
  u32x4_t rw[WPIECES];
  int cur_tap = 0, cur_c0 = -SBK;
  const float* apix[4] = {p.x, p.x, p.x, p.x};
  bool aok[4] = {false, false, false, false};
  const u32x4_t* wt = reinterpret_cast<const u32x4_t*>(p.w) + (int64_t)nblk * (TILE / 8) + nhalf * 256 + tid;
  const int64_t wt_step = (int64_t)p.nblks * (TILE / 8);

  auto fetch = [&]() __attribute__((always_inline)) {
    cur_c0 += SBK;
    if (cur_c0 >= p.kslabs * SBK) { cur_c0 = 0; ++cur_tap; }
    if (cur_c0 == 0) {                   // new tap: im2col coordinates and bounds once per Cin/32 slabs
      const int dy = cur_tap / p.KW, dx = cur_tap % p.KW;
      const int He = p.ups ? p.Hin * 2 : p.Hin, We = p.ups ? p.Win * 2 : p.Win;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int iy = ay[i] * p.stride + dy - p.pad_t, ix = ax[i] * p.stride + dx - p.pad_l;
        aok[i] = av[i] && iy >= 0 && iy < He && ix >= 0 && ix < We;
        if (p.ups) { iy >>= 1; ix >>= 1; }
        apix[i] = aok[i] ? p.x + (((int64_t)ab[i] * p.Hin + iy) * p.Win + ix) * p.ldx : p.x;
      }
    }
    const bool cok = cur_c0 + q * 4 < p.Cin;       // Cin % 4 == 0: a quad is inside or outside as a whole
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      kx[i] = aok[i] && cok;
      rx[i] = *reinterpret_cast<const float4*>(kx[i] ? apix[i] + cur_c0 + q * 4 : p.x);
    }
#pragma unroll
    for (int i = 0; i < WPIECES; ++i) rw[i] = wt[i * WSTEP];
    wt += wt_step;
  };
  auto stash = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < WPIECES; ++i) reinterpret_cast<u32x4_t*>(Ws)[i * WSTEP + tid] = rw[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4_t v = kx[i] ? f32x4_t{rx[i].x, rx[i].y, rx[i].z, rx[i].w} : f32x4_t{0.f, 0.f, 0.f, 0.f};
      store_split_quad(Xs, PLANE, (tid >> 3) + i * 32, q, v, ex);
    }
  };

  fetch();
  stash();
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    if (kt + 1 < nkt) fetch();           // next slab's global loads stay in flight under the MFMAs
    const bf16_t* wl = Ws + swz(wn * (NI * 16) + l16, g);
    const bf16_t* xl = Xs + swz(wm * 64 + l16, g);
    const bf16_t* xj[4] = {xl, xl + 16 * SBK, xl + 32 * SBK, xl + 48 * SBK};
    mma_slab<PLANE, NI>(wl, xj, acc);
    __syncthreads();                     // every wave is done reading this slab
    if (kt + 1 < nkt) stash();
    __syncthreads();
  }

  // epilogue: block (i, j): lane holds channels n..n+3 (n = 16 i + 4 g) of pixel 16 j + l16
  QuadStats qs[NI];
  float mx = 0.f;
#pragma unroll
  for (int i = 0; i < NI; ++i) qs[i] = QuadStats{0.0, 0.0};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int m = m0 + wm * 64 + j * 16 + l16;
    if (m >= p.M) continue;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int n = nblk * SBN + nhalf * 64 + wn * (NI * 16) + i * 16 + g * 4;
      if (n >= p.Cout) continue;
      const f32x4_t v = store_out_quad(acc[i][j], unscale, p.bias ? p.bias + n : nullptr,
                                       p.res ? p.res + (int64_t)m * p.ldres + n : nullptr, p.y + (int64_t)m * p.ldy + n, p.act);
      qs[i].add(v);
      mx = fmaxf(mx, quad_absmax(v));
    }
  }
  if (p.stats_out) {
    __syncthreads();
    flush_quad_stats<NI>(qs, mx, reinterpret_cast<double*>(Ws), wn * NI * 4, g, l16, tid, nblk * SBN + nhalf * 64, p.Cout, p.out_cpg,
                         p.stats_out + (int64_t)(m0 / hw) * (p.Cout / p.out_cpg) * 2, p.amax_out);
  }
}

// ------------------------------------------------------------------ 3x3 / stride 1 / pad 1: input patch resident in LDS
// The im2col form above re-reads (and re-splits) every input pixel once per tap.  Here a workgroup owns an 8 x 16
// block of output pixels, keeps the 10 x 18 input patch of one 32-channel slab in LDS as split planes, and runs all
// nine taps from it: activation traffic, split arithmetic and LDS writes drop 6.4x; only the weight tile changes per
// tap.  Optionally the patch load applies the preceding GroupNorm(+swish) (common_modules.py:19-27,308-335: every 3x3
// conv of a ResnetBlock reads swish(norm(x))), so the normalised tensor is never written to HBM.
constexpr int PT_H = 8, PT_W = 16, PP_W = PT_W + 2, PP_ROWS = (PT_H + 2) * PP_W;    // 180 patch pixels
constexpr int XPATCH = PP_ROWS * SBK;                                                // one plane of the patch

struct PatchArgs {
  const float* x; const bf16_t* w; const float* bias; const float* res; float* y;
  const float* w_amax; const float* x_amax;   // scale bounds, see SplitArgs
  int B, H, W, Cin, Cout, nblks, kslabs;
  int tiles_x, tiles_y, ntiles;
  const float2* mu_rstd;   // [B][G] mean / rstd of the input's GroupNorm, or null: x is used as is
  const float* gamma; const float* beta;
  int cpg, G, swish;
  double* stats_out;       // [B][Cout / out_cpg][2] fp64 sum / sum of squares of y (zeroed by the caller), or null
  int out_cpg;
  float* amax_out;         // max |y| (with stats_out)
};

// NI: 16-channel output blocks per wave.  4 = 128 output channels per workgroup; 2 = 64 (half of a weight tile's rows),
// for the small late layers whose 128-channel tiling leaves half the CUs without a workgroup.
template <bool GN, int NI>
__global__ __launch_bounds__(256, 2) void conv3x3_patch_kernel(PatchArgs p) {
  __shared__ __attribute__((aligned(16))) bf16_t Ws[TILE];
  __shared__ __attribute__((aligned(16))) bf16_t Xp[NPL * XPATCH];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave & 1, wm = wave >> 1;
  const int g = lane >> 4, l16 = lane & 15;
  const int ex = scale_exp(p.x_amax);
  const float unscale = __builtin_ldexpf(1.f, -(ex + scale_exp(p.w_amax)));
  const int per_xcd = gridDim.x >> 3;
  const int tile = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);     // contiguous band of tiles per XCD
  if (tile >= p.ntiles) return;
  const int nblk = NI == 4 ? blockIdx.y : blockIdx.y >> 1, nhalf = NI == 4 ? 0 : blockIdx.y & 1;
  const int tx = tile % p.tiles_x, ty = (tile / p.tiles_x) % p.tiles_y, b = tile / (p.tiles_x * p.tiles_y);
  const int y0 = ty * PT_H, x0 = tx * PT_W;

  // patch element e = tid + 256 i: patch pixel e >> 3, channel quad e & 7; the pixel's address is fixed for the kernel
  const int q = tid & 7;
  const float* ppix[6];
  bool pok[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int prow = (tid >> 3) + 32 * i;
    const int iy = y0 + prow / PP_W - 1, ix = x0 + prow % PP_W - 1;
    pok[i] = prow < PP_ROWS && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W && q * 4 < p.Cin;   // (Cin = 4: one quad of the slab)
    ppix[i] = pok[i] ? p.x + (((int64_t)b * p.H + iy) * p.W + ix) * p.Cin + q * 4 : p.x;
  }

  f32x4_t acc[NI][4];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  f32x4_t rx[6];
  // weight tile: NI == 4 all 128 rows of each plane (4 x 16 B per thread); NI == 2 this workgroup's 64 rows of each plane
  // (2 x 16 B per thread, one plane per piece), parked at rows 0..63 of the plane's LDS image
  constexpr int WPIECES = NI == 4 ? 4 : 2;
  constexpr int WSTEP = NI == 4 ? 256 : PLANE / 8;              // 16-byte units between a thread's pieces
  u32x4_t rw[WPIECES];
  const u32x4_t* wbase = reinterpret_cast<const u32x4_t*>(p.w) + (int64_t)nblk * (TILE / 8) + nhalf * 256 + tid;
  auto fetch_w = [&](int slab, int tap) __attribute__((always_inline)) {
    const u32x4_t* wt = wbase + ((int64_t)tap * p.kslabs + slab) * p.nblks * (TILE / 8);
#pragma unroll
    for (int i = 0; i < WPIECES; ++i) rw[i] = wt[i * WSTEP];
  };
  auto stash_w = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < WPIECES; ++i) reinterpret_cast<u32x4_t*>(Ws)[i * WSTEP + tid] = rw[i];
  };
  auto fetch_x = [&](int slab) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 6; ++i) rx[i] = *reinterpret_cast<const f32x4_t*>(pok[i] ? ppix[i] + slab * SBK : p.x);
  };
  auto stash_x = [&](int slab) __attribute__((always_inline)) {
    float mu = 0.f, rstd = 1.f;
    f32x4_t ga = {1.f, 1.f, 1.f, 1.f}, be = {0.f, 0.f, 0.f, 0.f};
    if constexpr (GN) {
      const int c = slab * SBK + q * 4;
      const float2 mr = p.mu_rstd[b * p.G + c / p.cpg];         // cpg >= 4: a quad never straddles groups
      mu = mr.x; rstd = mr.y;
      ga = *reinterpret_cast<const f32x4_t*>(p.gamma + c);
      be = *reinterpret_cast<const f32x4_t*>(p.beta + c);
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int prow = (tid >> 3) + 32 * i;
      if (prow >= PP_ROWS) continue;
      f32x4_t v = rx[i];
      if constexpr (GN) v = gn_swish_quad(v, mu, rstd, ga, be, p.swish);
      if (!pok[i]) v = f32x4_t{0.f, 0.f, 0.f, 0.f};             // the conv pads the NORMALISED tensor with zeros
      store_split_quad(Xp, XPATCH, prow, q, v, ex);
    }
  };

  fetch_x(0);
  fetch_w(0, 0);
  stash_x(0);
  stash_w();
  __syncthreads();
  const bf16_t* wl = Ws + swz(wn * (NI * 16) + l16, g);
  for (int slab = 0; slab < p.kslabs; ++slab) {
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
      const bool last_tap = tap == 8, more = !(last_tap && slab + 1 == p.kslabs);
      if (more) fetch_w(last_tap ? slab + 1 : slab, last_tap ? 0 : tap + 1);
      if (last_tap && more) fetch_x(slab + 1);
      const int dy = tap / 3, dx = tap - dy * 3;
      const bf16_t* xj[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) xj[j] = Xp + swz((wm * 4 + j + dy) * PP_W + l16 + dx, g);
      mma_slab<XPATCH, NI>(wl, xj, acc);
      __syncthreads();
      if (more) stash_w();
      if (last_tap && more) stash_x(slab + 1);
      __syncthreads();
    }
  }

  // epilogue: block (i, j): channels n..n+3 of output pixel (y0 + 4 wm + j, x0 + l16)
  const int ox = x0 + l16;
  QuadStats qs[NI];
  float mx = 0.f;
#pragma unroll
  for (int i = 0; i < NI; ++i) qs[i] = QuadStats{0.0, 0.0};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int oy = y0 + wm * 4 + j;
    if (oy >= p.H || ox >= p.W) continue;
    const int64_t m = ((int64_t)b * p.H + oy) * p.W + ox;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int n = nblk * SBN + nhalf * 64 + wn * (NI * 16) + i * 16 + g * 4;
      if (n >= p.Cout) continue;
      const f32x4_t v = store_out_quad(acc[i][j], unscale, p.bias ? p.bias + n : nullptr, p.res ? p.res + m * p.Cout + n : nullptr,
                                       p.y + m * p.Cout + n, 0);
      qs[i].add(v);
      mx = fmaxf(mx, quad_absmax(v));
    }
  }
  if (p.stats_out) {
    __syncthreads();                                         // the weight tile's LDS is free now
    flush_quad_stats<NI>(qs, mx, reinterpret_cast<double*>(Ws), wn * NI * 4, g, l16, tid, nblk * SBN + nhalf * 64, p.Cout, p.out_cpg,
                         p.stats_out + (int64_t)b * (p.Cout / p.out_cpg) * 2, p.amax_out);
  }
}

// Large-layer variant: 16 x 16 output pixels x 128 channels per workgroup of EIGHT waves (the weight tile of a tap is
// shared by twice the pixels), weight tiles in a three-slot LDS ring fetched two taps ahead by LDS-DMA -- the split
// image in HBM is the LDS image, so a tile is 16 linear 1 KB pieces, two per wave, no VGPR staging and one barrier
// per tap.  The three slots are separate arrays and the nine taps are unrolled so that every fragment read names a
// different object than the DMA in flight (otherwise hipcc drains vmcnt(0) before the first ds_read of every tap).
constexpr int QT_H = 16, QP_ROWS = (QT_H + 2) * PP_W;           // 324 patch pixels
constexpr int XQ = QP_ROWS * SBK;
// __syncthreads() also drains vmcnt(0); with DMA tiles deliberately left in flight the barrier is issued raw
#define RAW_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;


template <bool GN>
__global__ __launch_bounds__(512) void conv3x3_patch16_kernel(PatchArgs p) {
  __shared__ __attribute__((aligned(1024))) bf16_t W0[TILE];
  __shared__ __attribute__((aligned(1024))) bf16_t W1[TILE];
  __shared__ __attribute__((aligned(1024))) bf16_t W2[TILE];
  __shared__ __attribute__((aligned(16))) bf16_t Xp[NPL * XQ];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave & 1, wm = wave >> 1;
  const int g = lane >> 4, l16 = lane & 15;
  const int ex = scale_exp(p.x_amax);
  const float unscale = __builtin_ldexpf(1.f, -(ex + scale_exp(p.w_amax)));
  const int per_xcd = gridDim.x >> 3;
  const int tile = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if (tile >= p.ntiles) return;
  const int nblk = blockIdx.y;
  const int tx = tile % p.tiles_x, ty = (tile / p.tiles_x) % p.tiles_y, b = tile / (p.tiles_x * p.tiles_y);
  const int y0 = ty * QT_H, x0 = tx * PT_W;

  const int q = tid & 7;
  const float* ppix[6];
  bool pok[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int prow = (tid >> 3) + 64 * i;
    const int iy = y0 + prow / PP_W - 1, ix = x0 + prow % PP_W - 1;
    pok[i] = prow < QP_ROWS && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W && q * 4 < p.Cin;
    ppix[i] = pok[i] ? p.x + (((int64_t)b * p.H + iy) * p.W + ix) * p.Cin + q * 4 : p.x;
  }

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  f32x4_t rx[6];
  const char* wsrc = reinterpret_cast<const char*>(p.w) + (int64_t)nblk * (TILE * 2) + (wave * 2 * 64 + lane) * 16;
  auto dma_w = [&](int slab, int tap, bf16_t* slot) __attribute__((always_inline)) {
    const char* src = wsrc + ((int64_t)tap * p.kslabs + slab) * p.nblks * (TILE * 2);
    char* dst = reinterpret_cast<char*>(slot) + wave * 2 * 1024;
#pragma unroll
    for (int i = 0; i < 2; ++i) __builtin_amdgcn_global_load_lds((gptr_t)(src + i * 1024), (lptr_t)(dst + i * 1024), 16, 0, 0);
  };
  auto fetch_x = [&](int slab) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 6; ++i) rx[i] = *reinterpret_cast<const f32x4_t*>(pok[i] ? ppix[i] + slab * SBK : p.x);
  };
  auto stash_x = [&](int slab) __attribute__((always_inline)) {
    float mu = 0.f, rstd = 1.f;
    f32x4_t ga = {1.f, 1.f, 1.f, 1.f}, be = {0.f, 0.f, 0.f, 0.f};
    if constexpr (GN) {
      const int c = slab * SBK + q * 4;
      const float2 mr = p.mu_rstd[b * p.G + c / p.cpg];
      mu = mr.x; rstd = mr.y;
      ga = *reinterpret_cast<const f32x4_t*>(p.gamma + c);
      be = *reinterpret_cast<const f32x4_t*>(p.beta + c);
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int prow = (tid >> 3) + 64 * i;
      if (prow >= QP_ROWS) continue;
      f32x4_t v = rx[i];
      if constexpr (GN) v = gn_swish_quad(v, mu, rstd, ga, be, p.swish);
      if (!pok[i]) v = f32x4_t{0.f, 0.f, 0.f, 0.f};             // the conv pads the NORMALISED tensor with zeros
      store_split_quad(Xp, XQ, prow, q, v, ex);
    }
  };

  const int nkt = p.kslabs * 9;
  fetch_x(0);
  dma_w(0, 0, W0);
  dma_w(0, 1, W1);
  stash_x(0);
  asm volatile("s_waitcnt vmcnt(2)" ::: "memory");            // tap 0's tile; tap 1's (two pieces per wave) may still be in flight
  RAW_BARRIER();
  // Two wave groups (waves 0-3 / 4-7: one of each per SIMD) run every tap as L | M -- sixteen fragment reads + the DMA of the
  // tile two taps ahead | 48 MFMAs -- one phase apart, so that on every SIMD one wave's MFMAs cover the other's LDS reads
  // (the lock-stepped form had every wave read, then every wave multiply: matrix cores 0.45 busy).  The groups fall back into
  // step at the end of a slab, where the next 32-channel patch replaces this one.
  const int grp = wave >> 2;
  const int wrow = swz(wn * 64 + l16, g);
  for (int slab = 0; slab < p.kslabs; ++slab) {
    if (grp == 1) RAW_BARRIER();                                // group 1 starts one phase behind
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      bf16_t* cur = tap % 3 == 0 ? W0 : tap % 3 == 1 ? W1 : W2;
      bf16_t* two_ahead = (tap + 2) % 3 == 0 ? W0 : (tap + 2) % 3 == 1 ? W1 : W2;   // last read one tap ago (by either group)
      const bool has2 = slab * 9 + tap + 2 < nkt;
      const bool next_slab = tap == 8 && slab + 1 < p.kslabs;
      const int dy = tap / 3, dx = tap % 3;
      // ---------------- L phase
      h16x8_t w1[4], w2[4], x1[4], x2[4];
      const bf16_t* wl = cur + wrow;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        w1[i] = *reinterpret_cast<const h16x8_t*>(wl + i * 16 * SBK);
        w2[i] = *reinterpret_cast<const h16x8_t*>(wl + PLANE + i * 16 * SBK);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bf16_t* xr = Xp + swz((wm * 4 + j + dy) * PP_W + l16 + dx, g);
        x1[j] = *reinterpret_cast<const h16x8_t*>(xr);
        x2[j] = *reinterpret_cast<const h16x8_t*>(xr + XQ);
      }
      if (has2) dma_w(slab + (tap + 2 >= 9 ? 1 : 0), (tap + 2) % 9, two_ahead);
      if (next_slab) fetch_x(slab + 1);                         // six register loads behind the DMA: they stay in flight below
      // this wave's share of the next tap's tile (issued one tap ago) has landed; newer requests may stay in flight
      if (next_slab) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (has2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      RAW_BARRIER();
      // ---------------- M phase: three partial products per block, the small ones first (same order as mma_slab)
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[i], x2[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2[i], x1[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[i], x1[j], acc[i][j], 0, 0, 0);
        }
      __builtin_amdgcn_s_setprio(0);
      RAW_BARRIER();
    }
    if (grp == 0) RAW_BARRIER();                                // back in step: every wave is done with this slab's patch
    if (slab + 1 < p.kslabs) {
      stash_x(slab + 1);
      RAW_BARRIER();
    }
  }

  const int ox = x0 + l16;
  QuadStats qs[4];
  float mx = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) qs[i] = QuadStats{0.0, 0.0};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int oy = y0 + wm * 4 + j;
    if (oy >= p.H || ox >= p.W) continue;
    const int64_t m = ((int64_t)b * p.H + oy) * p.W + ox;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = nblk * SBN + wn * 64 + i * 16 + g * 4;
      if (n >= p.Cout) continue;
      const f32x4_t v = store_out_quad(acc[i][j], unscale, p.bias ? p.bias + n : nullptr, p.res ? p.res + m * p.Cout + n : nullptr,
                                       p.y + m * p.Cout + n, 0);
      qs[i].add(v);
      mx = fmaxf(mx, quad_absmax(v));
    }
  }
  if (p.stats_out) {
    __syncthreads();                                         // every DMA has landed (vmcnt(0) at the last tap) and been read
    flush_quad_stats<4>(qs, mx, reinterpret_cast<double*>(W0), wn * 16, g, l16, tid, nblk * SBN, p.Cout, p.out_cpg,
                        p.stats_out + (int64_t)b * (p.Cout / p.out_cpg) * 2, p.amax_out);
  }
}

// The 8-row variant with the 16-row kernel's weight path: tiles by LDS-DMA into a three-slot ring two taps ahead (separate arrays,
// taps unrolled: see above), fragments of a tap read BEFORE the tap's barrier, ONE barrier per tap -- the MFMAs of tap t run while
// the slower waves still read tap t + 1.  The register-staged form above waits out an L2 round trip per tap (fetch -> stash ->
// barrier -> read): 0.9 us per tap on the 512-channel layers at 16^2 / 32^2, where a workgroup runs 144 taps of 24-48 MFMAs.
template <bool GN, int NI>
__global__ __launch_bounds__(256, 2) void conv3x3_patch_dma_kernel(PatchArgs p) {
  __shared__ __attribute__((aligned(1024))) bf16_t W0[TILE];
  __shared__ __attribute__((aligned(1024))) bf16_t W1[TILE];
  __shared__ __attribute__((aligned(1024))) bf16_t W2[TILE];
  __shared__ __attribute__((aligned(16))) bf16_t Xp[NPL * XPATCH];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave & 1, wm = wave >> 1;
  const int g = lane >> 4, l16 = lane & 15;
  const int ex = scale_exp(p.x_amax);
  const float unscale = __builtin_ldexpf(1.f, -(ex + scale_exp(p.w_amax)));
  const int per_xcd = gridDim.x >> 3;
  const int tile = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if (tile >= p.ntiles) return;
  const int nblk = NI == 4 ? blockIdx.y : blockIdx.y >> 1, nhalf = NI == 4 ? 0 : blockIdx.y & 1;
  const int tx = tile % p.tiles_x, ty = (tile / p.tiles_x) % p.tiles_y, b = tile / (p.tiles_x * p.tiles_y);
  const int y0 = ty * PT_H, x0 = tx * PT_W;

  const int q = tid & 7;
  const float* ppix[6];
  bool pok[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int prow = (tid >> 3) + 32 * i;
    const int iy = y0 + prow / PP_W - 1, ix = x0 + prow % PP_W - 1;
    pok[i] = prow < PP_ROWS && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W && q * 4 < p.Cin;
    ppix[i] = pok[i] ? p.x + (((int64_t)b * p.H + iy) * p.W + ix) * p.Cin + q * 4 : p.x;
  }

  f32x4_t acc[NI][4];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  f32x4_t rx[6];
  // weight tile of a tap: NI == 4 the whole 16 KB image (4 one-KiB pieces per wave); NI == 2 this workgroup's 64 rows of each
  // plane (2 x 4 KB, one piece per wave and plane), parked at rows 0..63 of the plane's LDS image
  constexpr int WPIECES = NI == 4 ? 4 : 2;
  constexpr int WSTEP_B = NI == 4 ? 4096 : PLANE * 2;             // bytes between a wave's pieces (global and LDS alike)
  const char* wsrc = reinterpret_cast<const char*>(p.w) + (int64_t)nblk * (TILE * 2) + nhalf * 4096 + tid * 16;
  auto dma_w = [&](int slab, int tap, bf16_t* slot) __attribute__((always_inline)) {
    const char* src = wsrc + ((int64_t)tap * p.kslabs + slab) * p.nblks * (TILE * 2);
    char* dst = reinterpret_cast<char*>(slot) + wave * 1024;
#pragma unroll
    for (int i = 0; i < WPIECES; ++i) __builtin_amdgcn_global_load_lds((gptr_t)(src + i * WSTEP_B), (lptr_t)(dst + i * WSTEP_B), 16, 0, 0);
  };
  auto fetch_x = [&](int slab) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 6; ++i) rx[i] = *reinterpret_cast<const f32x4_t*>(pok[i] ? ppix[i] + slab * SBK : p.x);
  };
  auto stash_x = [&](int slab) __attribute__((always_inline)) {
    float mu = 0.f, rstd = 1.f;
    f32x4_t ga = {1.f, 1.f, 1.f, 1.f}, be = {0.f, 0.f, 0.f, 0.f};
    if constexpr (GN) {
      const int c = slab * SBK + q * 4;
      const float2 mr = p.mu_rstd[b * p.G + c / p.cpg];
      mu = mr.x; rstd = mr.y;
      ga = *reinterpret_cast<const f32x4_t*>(p.gamma + c);
      be = *reinterpret_cast<const f32x4_t*>(p.beta + c);
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int prow = (tid >> 3) + 32 * i;
      if (prow >= PP_ROWS) continue;
      f32x4_t v = rx[i];
      if constexpr (GN) v = gn_swish_quad(v, mu, rstd, ga, be, p.swish);
      if (!pok[i]) v = f32x4_t{0.f, 0.f, 0.f, 0.f};
      store_split_quad(Xp, XPATCH, prow, q, v, ex);
    }
  };

  const int nkt = p.kslabs * 9;
  fetch_x(0);
  dma_w(0, 0, W0);
  dma_w(0, 1, W1);
  stash_x(0);
  // tap 0's tile has landed (tap 1's pieces may still be in flight)
  if constexpr (WPIECES == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  RAW_BARRIER();
  const int wrow = swz(wn * (NI * 16) + l16, g);
  for (int slab = 0; slab < p.kslabs; ++slab) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      bf16_t* cur = tap % 3 == 0 ? W0 : tap % 3 == 1 ? W1 : W2;
      bf16_t* two_ahead = (tap + 2) % 3 == 0 ? W0 : (tap + 2) % 3 == 1 ? W1 : W2;   // last read one tap ago, before that tap's barrier
      const bool has2 = slab * 9 + tap + 2 < nkt;
      const bool next_slab = tap == 8 && slab + 1 < p.kslabs;
      const int dy = tap / 3, dx = tap % 3;
      h16x8_t w1[NI], w2[NI], x1[4], x2[4];
      const bf16_t* wl = cur + wrow;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        w1[i] = *reinterpret_cast<const h16x8_t*>(wl + i * 16 * SBK);
        w2[i] = *reinterpret_cast<const h16x8_t*>(wl + PLANE + i * 16 * SBK);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bf16_t* xr = Xp + swz((wm * 4 + j + dy) * PP_W + l16 + dx, g);
        x1[j] = *reinterpret_cast<const h16x8_t*>(xr);
        x2[j] = *reinterpret_cast<const h16x8_t*>(xr + XPATCH);
      }
      if (has2) dma_w(slab + (tap + 2 >= 9 ? 1 : 0), (tap + 2) % 9, two_ahead);
      if (next_slab) fetch_x(slab + 1);                         // six register loads behind the DMA
      // the NEXT tap's tile (issued one tap ago) has landed; the pieces issued just now (and the patch loads) may stay in flight
      if (next_slab) { if constexpr (WPIECES == 4) asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
      else if (has2) { if constexpr (WPIECES == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      RAW_BARRIER();                                            // everyone's fragments of this tap are in registers, the next tile is complete
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[i], x2[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2[i], x1[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[i], x1[j], acc[i][j], 0, 0, 0);
        }
    }
    if (slab + 1 < p.kslabs) {
      stash_x(slab + 1);                                        // (every wave's reads of this slab's patch came before the last barrier)
      RAW_BARRIER();
    }
  }

  const int ox = x0 + l16;
  QuadStats qs[NI];
  float mx = 0.f;
#pragma unroll
  for (int i = 0; i < NI; ++i) qs[i] = QuadStats{0.0, 0.0};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int oy = y0 + wm * 4 + j;
    if (oy >= p.H || ox >= p.W) continue;
    const int64_t m = ((int64_t)b * p.H + oy) * p.W + ox;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int n = nblk * SBN + nhalf * 64 + wn * (NI * 16) + i * 16 + g * 4;
      if (n >= p.Cout) continue;
      const f32x4_t v = store_out_quad(acc[i][j], unscale, p.bias ? p.bias + n : nullptr, p.res ? p.res + m * p.Cout + n : nullptr,
                                       p.y + m * p.Cout + n, 0);
      qs[i].add(v);
      mx = fmaxf(mx, quad_absmax(v));
    }
  }
  if (p.stats_out) {
    __syncthreads();
    flush_quad_stats<NI>(qs, mx, reinterpret_cast<double*>(W0), wn * NI * 4, g, l16, tid, nblk * SBN + nhalf * 64, p.Cout, p.out_cpg,
                         p.stats_out + (int64_t)b * (p.Cout / p.out_cpg) * 2, p.amax_out);
  }
}

// packed fp32 weights [taps][Cin][cout_pad] -> per (tap, slab, cout block) tile: 2 planes x [128 cout][32 cin] fp16 of
// the weights scaled by 2^ew (ew from *amax, the tensor's max|w| computed just before), in the swizzled LDS image.
// One thread per 16-byte chunk of the output.
__global__ __launch_bounds__(256) void conv_split_weights_kernel(const float* __restrict__ wp, bf16_t* __restrict__ out,
                                                                 const float* __restrict__ amax, int taps, int Cin, int cout_pad) {
  const int kslabs = (Cin + SBK - 1) / SBK, nblks = cout_pad / SBN;
  const int64_t chunks = (int64_t)taps * kslabs * nblks * (TILE / 8);
  const int ew = scale_exp(amax);
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < chunks; idx += (int64_t)gridDim.x * blockDim.x) {
    const int within = (int)(idx % (TILE / 8));
    const int64_t tile = idx / (TILE / 8);
    const int plane = within / (PLANE / 8);
    const int rc = within % (PLANE / 8);
    const int r = rc >> 2, chunk = (rc & 3) ^ swz_q(r);              // stored position -> logical chunk
    const int nb = (int)(tile % nblks);
    const int ks = (int)((tile / nblks) % kslabs);
    const int tap = (int)(tile / ((int64_t)nblks * kslabs));
    bf16_t o[8];
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
      const int c = ks * SBK + chunk * 8 + e;
      const float a = c < Cin ? wp[((int64_t)tap * Cin + c) * cout_pad + nb * SBN + r] : 0.f;
      const float b = c + 1 < Cin ? wp[((int64_t)tap * Cin + c + 1) * cout_pad + nb * SBN + r] : 0.f;
      uint32_t p1, p2;
      split2_pair(__builtin_ldexpf(a, ew), __builtin_ldexpf(b, ew), p1, p2);
      const uint32_t pk = plane == 0 ? p1 : p2;
      o[e] = (bf16_t)(pk & 0xffffu);
      o[e + 1] = (bf16_t)(pk >> 16);
    }
    *reinterpret_cast<uint4*>(out + idx * 8) = *reinterpret_cast<const uint4*>(o);
  }
}

// max|x| over rows x cols fp32 (row stride ld), as an order-preserving unsigned max of the float bits into *out
// (cleared by a memset node ahead of the launch)
__global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ x, int64_t rows, int64_t cols, int64_t ld,
                                                   float* __restrict__ out, int vec) {
  __shared__ float red[4];
  float m = 0.f;
  const int64_t cv = vec ? (cols & ~3LL) : 0;           // columns covered by 16-byte loads (aligned rows only)
  if (cv) {
    const int64_t c4 = cv >> 2, total4 = rows * c4;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
      const int64_t r = i / c4, c = (i % c4) * 4;
      const float4 v = *reinterpret_cast<const float4*>(x + r * ld + c);
      m = fmaxf(fmaxf(fmaxf(m, fabsf(v.x)), fmaxf(fabsf(v.y), fabsf(v.z))), fabsf(v.w));
    }
  }
  if (cv < cols) {
    const int64_t tail = cols - cv, totalt = rows * tail;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < totalt; i += (int64_t)gridDim.x * blockDim.x)
      m = fmaxf(m, fabsf(x[(i / tail) * ld + cv + i % tail]));
  }
  m = block_max<4>(m, red);
  if (threadIdx.x == 0 && m > 0.f) atomicMax(reinterpret_cast<unsigned int*>(out), __float_as_uint(m));
}

int launch_amax(const float* x, int64_t rows, int64_t cols, int64_t ld, float* out, hipStream_t st, bool zeroed = false) {
  if (!zeroed) UG_HIP(hipMemsetAsync(out, 0, sizeof(float), st));
  int64_t g = (rows * cols / 4 + 255) / 256;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  const int vec = ug_aligned16(x) && ld % 4 == 0;
  hipLaunchKernelGGL(amax_kernel, dim3((unsigned)g), dim3(256), 0, st, x, rows, cols, ld, out, vec);
  UG_CHECK_LAUNCH("ug_amax_f32");
  return UG_OK;
}

// 64-channel workgroups when the 128-channel tiling would not give every CU two workgroups
void launch_split(const SplitArgs& a, int64_t M, int64_t N, hipStream_t st) {
  const unsigned mt = (unsigned)(((M + SBM - 1) / SBM + 7) / 8 * 8);
  const int64_t wgs = ((M + SBM - 1) / SBM) * ((N + SBN - 1) / SBN);
  if (wgs < 512) hipLaunchKernelGGL(conv_split_kernel<2>, dim3(mt, (unsigned)((N + 63) / 64)), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(conv_split_kernel<4>, dim3(mt, (unsigned)((N + SBN - 1) / SBN)), dim3(256), 0, st, a);
}

}  // namespace

// ------------------------------------------------------------------------------------ C ABI
// 16-bit elements of a split weight buffer: the tiles, then 8 elements whose first 4 bytes hold max|w| (fp32)
static inline int64_t split_tile_elems(int taps, int Cin, int cout_pad) {
  return (int64_t)taps * ((Cin + SBK - 1) / SBK) * (cout_pad / SBN) * TILE;
}

extern "C" int ug_amax_f32(const float* x, int64_t rows, int64_t cols, int64_t ld, float* out_amax, hipStream_t st) {
  UG_REQUIRE(x && out_amax && rows > 0 && cols > 0 && ld >= cols, "ug_amax_f32: bad args");
  return launch_amax(x, rows, cols, ld, out_amax, st);
}

// ... into a slot the CALLER keeps zeroed (a pool cleared once): no memset node ahead of the launch -- the SigLIP tower measures 135 bounds
// per forward and the 4-byte memset + its command boundary cost more than the pass over the tensor (34.6 us per bound)
extern "C" int ug_amax_f32_into_zeroed(const float* x, int64_t rows, int64_t cols, int64_t ld, float* out_amax, hipStream_t st) {
  UG_REQUIRE(x && out_amax && rows > 0 && cols > 0 && ld >= cols, "ug_amax_f32_into_zeroed: bad args");
  return launch_amax(x, rows, cols, ld, out_amax, st, true);
}

extern "C" int ug_conv_split_weights(const float* w_packed, uint16_t* w_split, int taps, int Cin, int cout_pad,
                                     hipStream_t st) {
  UG_REQUIRE(w_packed && w_split && taps > 0 && Cin > 0 && cout_pad > 0 && cout_pad % SBN == 0,
             "ug_conv_split_weights: needs cout_pad %% 128 == 0 (Cin=%d cout_pad=%d)", Cin, cout_pad);
  UG_REQUIRE(ug_aligned16(w_split) && ug_aligned16(w_packed), "ug_conv_split_weights: buffers must be 16-byte aligned");
  const int64_t elems = split_tile_elems(taps, Cin, cout_pad);
  float* amax = reinterpret_cast<float*>(w_split + elems);
  if (int rc = launch_amax(w_packed, 1, (int64_t)taps * Cin * cout_pad, (int64_t)taps * Cin * cout_pad, amax, st)) return rc;
  const int64_t chunks = elems / 8;
  int64_t g = (chunks + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(conv_split_weights_kernel, dim3((unsigned)g), dim3(256), 0, st, w_packed, (bf16_t*)w_split, amax, taps, Cin,
                     cout_pad);
  UG_CHECK_LAUNCH("ug_conv_split_weights");
  return UG_OK;
}

extern "C" int ug_conv2d_split(const float* x, const float* x_amax, const uint16_t* w_split, const float* bias, const float* residual,
                               float* y, int64_t B, int Hin, int Win, int Cin, int Cout, int cout_pad, int ksize, int stride,
                               int pad_top, int pad_left, int Hout, int Wout, int upsample2x, double* out_stats, int out_groups,
                               hipStream_t st) {
  UG_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && ksize >= 1 && ksize <= 16, "ug_conv2d_split: bad shape");
  UG_REQUIRE(!out_stats || (out_groups > 0 && Cout % out_groups == 0 && (Cout / out_groups) % 4 == 0 && (Hout * Wout) % SBM == 0 &&
                            ((uintptr_t)out_stats & 7) == 0),
             "ug_conv2d_split: output statistics need channels-per-group %% 4 == 0 and Hout * Wout %% 128 == 0 (Cout=%d groups=%d)",
             Cout, out_groups);
  UG_REQUIRE(Cin % 4 == 0 && Cout % 4 == 0 && cout_pad % SBN == 0 && cout_pad >= Cout,
             "ug_conv2d_split: needs Cin %% 4 == 0, Cout %% 4 == 0, cout_pad %% 128 == 0 (Cin=%d Cout=%d cout_pad=%d)", Cin,
             Cout, cout_pad);
  UG_REQUIRE(x && y && w_split && ug_aligned16(x) && ug_aligned16(y) && ug_aligned16(w_split) &&
                 (!bias || ug_aligned16(bias)) && (!residual || ug_aligned16(residual)),
             "ug_conv2d_split: pointers must be 16-byte aligned");
  const int64_t M = B * Hout * Wout;
  UG_REQUIRE(M < (1LL << 31), "ug_conv2d_split: too many output pixels");
  SplitArgs a{};
  a.x = x; a.w = (const bf16_t*)w_split; a.bias = bias; a.res = residual; a.y = y;
  a.x_amax = x_amax; a.w_amax = reinterpret_cast<const float*>(w_split + split_tile_elems(ksize * ksize, Cin, cout_pad));
  a.B = (int)B; a.Hin = Hin; a.Win = Win; a.Cin = Cin; a.Hout = Hout; a.Wout = Wout; a.Cout = Cout;
  a.KH = ksize; a.KW = ksize; a.stride = stride; a.pad_t = pad_top; a.pad_l = pad_left; a.ups = upsample2x;
  a.nblks = cout_pad / SBN; a.M = (int)M; a.kslabs = (Cin + SBK - 1) / SBK;
  a.ldx = Cin; a.ldy = Cout; a.ldres = Cout;
  if (out_stats) {
    a.stats_out = out_stats; a.out_cpg = Cout / out_groups;
    a.amax_out = reinterpret_cast<float*>(out_stats + 2 * B * out_groups);
  }
  launch_split(a, M, Cout, st);
  UG_CHECK_LAUNCH("ug_conv2d_split");
  return UG_OK;
}

extern "C" int ug_linear_split(const float* x, int64_t ldx, const float* x_amax, const uint16_t* w_split, const float* bias,
                               const float* residual, int64_t ldres, float* y, int64_t ldy, int64_t M, int64_t N, int64_t K,
                               int n_pad, int act, hipStream_t st) {
  UG_REQUIRE(M > 0 && N > 0 && K > 0 && (act == 0 || act == 1) && M < (1LL << 31), "ug_linear_split: bad args");
  UG_REQUIRE(K % 4 == 0 && N % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && (!residual || ldres % 4 == 0) && n_pad % SBN == 0 &&
                 n_pad >= N,
             "ug_linear_split: needs K, N and the row strides %% 4 == 0 and n_pad %% 128 == 0 (K=%lld N=%lld n_pad=%d)",
             (long long)K, (long long)N, n_pad);
  UG_REQUIRE(x && y && w_split && ug_aligned16(x) && ug_aligned16(y) && ug_aligned16(w_split) &&
                 (!bias || ug_aligned16(bias)) && (!residual || ug_aligned16(residual)),
             "ug_linear_split: pointers must be 16-byte aligned");
  SplitArgs a{};
  a.x = x; a.w = (const bf16_t*)w_split; a.bias = bias; a.res = residual; a.y = y;
  a.x_amax = x_amax; a.w_amax = reinterpret_cast<const float*>(w_split + split_tile_elems(1, (int)K, n_pad));
  a.B = (int)M; a.Hin = a.Win = a.Hout = a.Wout = 1; a.Cin = (int)K; a.Cout = (int)N;
  a.KH = a.KW = 1; a.stride = 1;
  a.nblks = n_pad / SBN; a.M = (int)M; a.kslabs = (int)((K + SBK - 1) / SBK);
  a.ldx = ldx; a.ldy = ldy; a.ldres = ldres; a.act = act;
  launch_split(a, M, N, st);
  UG_CHECK_LAUNCH("ug_linear_split");
  return UG_OK;
}

extern "C" int ug_conv3x3_split(const float* x, const float* x_amax, const uint16_t* w_split, const float* bias, const float* residual,
                                float* y, int64_t B, int H, int W, int Cin, int Cout, int cout_pad, const float* gn_mu_rstd,
                                const float* gn_gamma, const float* gn_beta, int gn_groups, int gn_swish, double* out_stats,
                                int out_groups, hipStream_t st) {
  UG_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "ug_conv3x3_split: bad shape");
  UG_REQUIRE(!out_stats || (out_groups > 0 && Cout % out_groups == 0 && (Cout / out_groups) % 4 == 0 && ((uintptr_t)out_stats & 7) == 0),
             "ug_conv3x3_split: output statistics need channels-per-group %% 4 == 0 (Cout=%d groups=%d)", Cout, out_groups);
  // Cin = 4 (RGB + a zero channel, 16-byte pixels): one 32-channel slab whose other 28 channels read as zero -- the weights are
  // packed for Cin padded to 32 (same tile count); no GroupNorm on the load path then
  UG_REQUIRE((Cin % SBK == 0 || (Cin == 4 && !gn_mu_rstd)) && Cout % 4 == 0 && cout_pad % SBN == 0 && cout_pad >= Cout,
             "ug_conv3x3_split: needs Cin %% 32 == 0 (or 4), Cout %% 4 == 0, cout_pad %% 128 == 0 (Cin=%d Cout=%d cout_pad=%d)", Cin,
             Cout, cout_pad);
  UG_REQUIRE(x && y && w_split && ug_aligned16(x) && ug_aligned16(y) && ug_aligned16(w_split) &&
                 (!bias || ug_aligned16(bias)) && (!residual || ug_aligned16(residual)),
             "ug_conv3x3_split: pointers must be 16-byte aligned");
  PatchArgs a{};
  a.x = x; a.w = (const bf16_t*)w_split; a.bias = bias; a.res = residual; a.y = y;
  a.x_amax = x_amax; a.w_amax = reinterpret_cast<const float*>(w_split + split_tile_elems(9, Cin, cout_pad));
  a.B = (int)B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.nblks = cout_pad / SBN; a.kslabs = (Cin + SBK - 1) / SBK;
  if (out_stats) {
    a.stats_out = out_stats; a.out_cpg = Cout / out_groups;
    a.amax_out = reinterpret_cast<float*>(out_stats + 2 * B * out_groups);
  }
  // 16-row tiles (eight waves, DMA-fed weights) when they still give every CU a workgroup
  const int nb_n = (Cout + SBN - 1) / SBN;
  const int64_t big_tiles = B * ((W + PT_W - 1) / PT_W) * ((H + QT_H - 1) / QT_H);
  // (a one-slab layer -- conv_in, nine taps per workgroup, write-bound: 537 MB at 256^2 -- measures the same on either variant:
  // 303 us on this one, 322 us on the four-wave one)
  // (>= one workgroup per CU: the 512 -> 512 layers at 32^2 -- 256 such workgroups for 16 images -- run 227 -> ~150 us on it;
  // measured per threshold, get_code on 16 images: 512: 18.41 ms, 256: 18.07, 128: 18.06, 64: 18.49)
  static const int big_min = [] { const char* e = getenv("UNIGEN_CONV_BIG_MIN"); return e ? atoi(e) : 256; }();
  const bool big = big_tiles * nb_n >= big_min;
  a.tiles_x = (W + PT_W - 1) / PT_W; a.tiles_y = big ? (H + QT_H - 1) / QT_H : (H + PT_H - 1) / PT_H;
  const int64_t ntiles = B * a.tiles_x * a.tiles_y;
  UG_REQUIRE(ntiles < (1LL << 30) && B * H * W < (1LL << 31), "ug_conv3x3_split: too many output pixels");
  a.ntiles = (int)ntiles;
  // 64-channel workgroups when the 128-channel tiling of the 8-row variant would leave CUs without work
  const bool half = !big && ntiles * nb_n < 512;
  dim3 grid((unsigned)((ntiles + 7) / 8 * 8), (unsigned)(half ? (Cout + 63) / 64 : nb_n));
  const dim3 block(big ? 512 : 256);
  static const int dma8 = [] { const char* e = getenv("UNIGEN_CONV_DMA8"); return e ? atoi(e) : 1; }();
  if (gn_mu_rstd) {
    UG_REQUIRE(gn_gamma && gn_beta && gn_groups > 0 && Cin % gn_groups == 0 && (Cin / gn_groups) % 4 == 0 &&
                   ug_aligned16(gn_gamma) && ug_aligned16(gn_beta) && ((uintptr_t)gn_mu_rstd & 7) == 0,
               "ug_conv3x3_split: fused GroupNorm needs gamma/beta and channels-per-group %% 4 == 0 (Cin=%d groups=%d)", Cin,
               gn_groups);
    a.mu_rstd = reinterpret_cast<const float2*>(gn_mu_rstd); a.gamma = gn_gamma; a.beta = gn_beta;
    a.G = gn_groups; a.cpg = Cin / gn_groups; a.swish = gn_swish;
    if (big) hipLaunchKernelGGL(conv3x3_patch16_kernel<true>, grid, block, 0, st, a);
    else if (half && dma8) hipLaunchKernelGGL((conv3x3_patch_dma_kernel<true, 2>), grid, block, 0, st, a);
    else if (half) hipLaunchKernelGGL((conv3x3_patch_kernel<true, 2>), grid, block, 0, st, a);
    else if (dma8) hipLaunchKernelGGL((conv3x3_patch_dma_kernel<true, 4>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((conv3x3_patch_kernel<true, 4>), grid, block, 0, st, a);
  } else {
    if (big) hipLaunchKernelGGL(conv3x3_patch16_kernel<false>, grid, block, 0, st, a);
    else if (half && dma8) hipLaunchKernelGGL((conv3x3_patch_dma_kernel<false, 2>), grid, block, 0, st, a);
    else if (half) hipLaunchKernelGGL((conv3x3_patch_kernel<false, 2>), grid, block, 0, st, a);
    else if (dma8) hipLaunchKernelGGL((conv3x3_patch_dma_kernel<false, 4>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((conv3x3_patch_kernel<false, 4>), grid, block, 0, st, a);
  }
  UG_CHECK_LAUNCH("ug_conv3x3_split");
  return UG_OK;
}
