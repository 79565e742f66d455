// MAGVITv2 tokenizer convolutions at fp32 accuracy on the bf16 matrix cores (reference: common_modules.py:19-360,
// magvitv2.py:57-200 -- every nn.Conv2d of VQGANEncoder / VQGANDecoder runs in fp32 there).
//
// gfx950's bf16 MFMA runs 16x the rate of its fp32 MFMA, so an fp32 product is rebuilt from bf16 pieces:
// every fp32 value is split into three bf16 terms a = a1 + a2 + a3 (each the round-to-nearest bf16 of the
// remainder; 3 x 8 significand bits carry all 24), and a.b is summed from the six partial products whose weight
// is >= 2^-16 of the full product:  a1b1 + a1b2 + a2b1 + a1b3 + a2b2 + a3b1.  Each bf16 x bf16 product is exact
// in the fp32 accumulator; the three dropped terms are <= 3 * 2^-26 |a.b|, below the fp32 rounding of the sum
// itself.  Six bf16 MFMAs cost 6/16 of one fp32 MFMA, so the bound moves from 157 TF/s to ~410 TF/s equivalent.
//
// Implicit GEMM  y[m, n] = sum_{tap, c} x[pix(m, tap), c] * w[tap][c][n]  with the WEIGHTS as the MFMA A operand
// (rows = cout) and the PIXELS as the B operand, so that each lane of a 16x16 result block holds four consecutive
// output channels of one pixel (one float4 store, NHWC).  Weights are split once at pack time
// (ug_conv_split_weights) into the exact LDS image of each (tap, 32-channel slab, 128-cout block) tile; activations
// are split on the way from registers to LDS.
#include "common.h"
#include "unigen_hip.h"

namespace {

constexpr int SBM = 128;                 // output pixels per workgroup
constexpr int SBN = 128;                 // output channels per workgroup
constexpr int SBK = 32;                  // contraction slab = one 16x16x32 MFMA step
constexpr int PLANE = SBN * SBK;         // bf16 elements of one plane of one tile (128 rows x 32 k)
constexpr int TILE = 3 * PLANE;          // three planes
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;

struct SplitArgs {
  const float* x;          // [B, Hin, Win, Cin] fp32 NHWC
  const bf16_t* w;         // split tiles, see ug_conv_split_weights
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
};

__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) { return pack_bf2(lo, hi); }   // v_cvt_pk_bf16_f32
// two fp32 values -> three packed bf16 pairs
__device__ __forceinline__ void split3_pair(float a, float b, uint32_t& p1, uint32_t& p2, uint32_t& p3) {
  p1 = cvt_pk_bf16(a, b);
  const float ra = a - __uint_as_float(p1 << 16), rb = b - __uint_as_float(p1 & 0xffff0000u);
  p2 = cvt_pk_bf16(ra, rb);
  const float sa = ra - __uint_as_float(p2 << 16), sb = rb - __uint_as_float(p2 & 0xffff0000u);
  p3 = cvt_pk_bf16(sa, sb);
}
// LDS image of one plane: row r (64 B), 16-byte chunk c (8 k) stored at chunk c ^ ((r >> 2) & 3): the 16 rows x 1 chunk
// a ds_read_b128 fragment fetch touches land on 16 distinct chunks of a 256-byte bank row
__device__ __forceinline__ int swz(int r, int chunk) { return r * SBK + ((chunk ^ ((r >> 2) & 3)) << 3); }

// four fp32 values of one pixel (channel quad `q` of LDS row `row`) -> the three split planes
__device__ __forceinline__ void store_split_quad(bf16_t* planes, int plane_stride, int row, int q, f32x4_t v) {
  uint32_t a1, a2, a3, b1, b2, b3;
  split3_pair(v[0], v[1], a1, a2, a3);
  split3_pair(v[2], v[3], b1, b2, b3);
  const int off = swz(row, q >> 1) + (q & 1) * 4;
  *reinterpret_cast<uint2*>(planes + off) = make_uint2(a1, b1);
  *reinterpret_cast<uint2*>(planes + plane_stride + off) = make_uint2(a2, b2);
  *reinterpret_cast<uint2*>(planes + 2 * plane_stride + off) = make_uint2(a3, b3);
}

// result block epilogue: four consecutive output channels of one pixel.  y = act(acc + bias) + residual
__device__ __forceinline__ void store_out_quad(f32x4_t a, const float* bias, const float* res, float* y, int act) {
  float v[4] = {a[0], a[1], a[2], a[3]};
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

// One 32-deep contraction slab of a wave's 64 x 64 result (4 x 4 MFMA blocks): six partial products per block, the
// smallest first.  wl: this lane's row of the weight plane-0 image (blocks 16 rows apart); xj[j]: this lane's row of
// pixel block j in plane 0; planes are XPLANE (pixels) / PLANE (weights) elements apart.
template <int XPLANE, int NI = 4>
__device__ __forceinline__ void mma_slab(const bf16_t* wl, const bf16_t* const (&xj)[4], f32x4_t (&acc)[NI][4]) {
  bf16x8_t w1[NI], w2[NI], x1[4], x2[4], tx[4], tw[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) w1[i] = *reinterpret_cast<const bf16x8_t*>(wl + i * 16 * SBK);
#pragma unroll
  for (int j = 0; j < 4; ++j) x1[j] = *reinterpret_cast<const bf16x8_t*>(xj[j]);
#pragma unroll
  for (int j = 0; j < 4; ++j) tx[j] = *reinterpret_cast<const bf16x8_t*>(xj[j] + 2 * XPLANE);   // x3
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[i], tx[j], acc[i][j], 0, 0, 0);
#pragma unroll
  for (int i = 0; i < NI; ++i) tw[i] = *reinterpret_cast<const bf16x8_t*>(wl + 2 * PLANE + i * 16 * SBK);   // w3
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tw[i], x1[j], acc[i][j], 0, 0, 0);
#pragma unroll
  for (int i = 0; i < NI; ++i) w2[i] = *reinterpret_cast<const bf16x8_t*>(wl + PLANE + i * 16 * SBK);
#pragma unroll
  for (int j = 0; j < 4; ++j) x2[j] = *reinterpret_cast<const bf16x8_t*>(xj[j] + XPLANE);
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2[i], x2[j], acc[i][j], 0, 0, 0);
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[i], x2[j], acc[i][j], 0, 0, 0);
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2[i], x1[j], acc[i][j], 0, 0, 0);
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[i], x1[j], acc[i][j], 0, 0, 0);
    }
}

// NI: 16-channel output blocks per wave (4: 128 output channels per workgroup; 2: 64, for outputs whose 128-wide tiling
// leaves CUs idle -- SigLIP's 1152-wide projections at four images, the tokenizer's 1x1 convs at 16 x 16)
template <int NI>
__global__ __launch_bounds__(256, 2) void conv_split3_kernel(SplitArgs p) {
  __shared__ __attribute__((aligned(16))) bf16_t Ws[TILE];
  __shared__ __attribute__((aligned(16))) bf16_t Xs[TILE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave & 1, wm = wave >> 1;
  const int g = lane >> 4, l16 = lane & 15;
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
  constexpr int WPIECES = NI == 4 ? 6 : 3;                  // NI == 2: this workgroup's 64 rows of each weight plane
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
      store_split_quad(Xs, PLANE, (tid >> 3) + i * 32, q, v);
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
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int m = m0 + wm * 64 + j * 16 + l16;
    if (m >= p.M) continue;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int n = nblk * SBN + nhalf * 64 + wn * (NI * 16) + i * 16 + g * 4;
      if (n >= p.Cout) continue;
      store_out_quad(acc[i][j], p.bias ? p.bias + n : nullptr, p.res ? p.res + (int64_t)m * p.ldres + n : nullptr,
                     p.y + (int64_t)m * p.ldy + n, p.act);
    }
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
  int B, H, W, Cin, Cout, nblks, kslabs;
  int tiles_x, tiles_y, ntiles;
  const float2* mu_rstd;   // [B][G] mean / rstd of the input's GroupNorm, or null: x is used as is
  const float* gamma; const float* beta;
  int cpg, G, swish;
};

// NI: 16-channel output blocks per wave.  4 = 128 output channels per workgroup; 2 = 64 (half of a weight tile's rows),
// for the small late layers whose 128-channel tiling leaves half the CUs without a workgroup.
template <bool GN, int NI>
__global__ __launch_bounds__(256, 2) void conv3x3_patch_kernel(PatchArgs p) {
  __shared__ __attribute__((aligned(16))) bf16_t Ws[TILE];
  __shared__ __attribute__((aligned(16))) bf16_t Xp[3 * XPATCH];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave & 1, wm = wave >> 1;
  const int g = lane >> 4, l16 = lane & 15;
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
    pok[i] = prow < PP_ROWS && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
    ppix[i] = pok[i] ? p.x + (((int64_t)b * p.H + iy) * p.W + ix) * p.Cin + q * 4 : p.x;
  }

  f32x4_t acc[NI][4];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  f32x4_t rx[6];
  // weight tile: NI == 4 all 128 rows of each plane (6 x 16 B per thread); NI == 2 this workgroup's 64 rows of each plane
  // (3 x 16 B per thread, one plane per piece), parked at rows 0..63 of the plane's LDS image
  constexpr int WPIECES = NI == 4 ? 6 : 3;
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
      store_split_quad(Xp, XPATCH, prow, q, v);
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
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int oy = y0 + wm * 4 + j;
    if (oy >= p.H || ox >= p.W) continue;
    const int64_t m = ((int64_t)b * p.H + oy) * p.W + ox;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int n = nblk * SBN + nhalf * 64 + wn * (NI * 16) + i * 16 + g * 4;
      if (n >= p.Cout) continue;
      store_out_quad(acc[i][j], p.bias ? p.bias + n : nullptr, p.res ? p.res + m * p.Cout + n : nullptr,
                     p.y + m * p.Cout + n, 0);
    }
  }
}

// Large-layer variant: 16 x 16 output pixels x 128 channels per workgroup of EIGHT waves (the weight tile of a tap is
// shared by twice the pixels), weight tiles in a three-slot LDS ring fetched two taps ahead by LDS-DMA -- the split
// image in HBM is the LDS image, so a tile is 24 linear 1 KB pieces, three per wave, no VGPR staging and one barrier
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
  __shared__ __attribute__((aligned(16))) bf16_t Xp[3 * XQ];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave & 1, wm = wave >> 1;
  const int g = lane >> 4, l16 = lane & 15;
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
    pok[i] = prow < QP_ROWS && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
    ppix[i] = pok[i] ? p.x + (((int64_t)b * p.H + iy) * p.W + ix) * p.Cin + q * 4 : p.x;
  }

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  f32x4_t rx[6];
  const char* wsrc = reinterpret_cast<const char*>(p.w) + (int64_t)nblk * (TILE * 2) + (wave * 3 * 64 + lane) * 16;
  auto dma_w = [&](int slab, int tap, bf16_t* slot) __attribute__((always_inline)) {
    const char* src = wsrc + ((int64_t)tap * p.kslabs + slab) * p.nblks * (TILE * 2);
    char* dst = reinterpret_cast<char*>(slot) + wave * 3 * 1024;
#pragma unroll
    for (int i = 0; i < 3; ++i) __builtin_amdgcn_global_load_lds((gptr_t)(src + i * 1024), (lptr_t)(dst + i * 1024), 16, 0, 0);
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
      store_split_quad(Xp, XQ, prow, q, v);
    }
  };

  const int nkt = p.kslabs * 9;
  fetch_x(0);
  dma_w(0, 0, W0);
  dma_w(0, 1, W1);
  stash_x(0);
  asm volatile("s_waitcnt vmcnt(3)" ::: "memory");            // tap 0's tile; tap 1's may still be in flight
  RAW_BARRIER();
  const int wrow = swz(wn * 64 + l16, g);
  for (int slab = 0; slab < p.kslabs; ++slab) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      bf16_t* cur = tap % 3 == 0 ? W0 : tap % 3 == 1 ? W1 : W2;
      bf16_t* two_ahead = (tap + 2) % 3 == 0 ? W0 : (tap + 2) % 3 == 1 ? W1 : W2;   // last read one tap ago
      const bool has2 = slab * 9 + tap + 2 < nkt;
      const bool next_slab = tap == 8 && slab + 1 < p.kslabs;
      if (has2) dma_w(slab + (tap + 2 >= 9 ? 1 : 0), (tap + 2) % 9, two_ahead);
      if (next_slab) fetch_x(slab + 1);
      const int dy = tap / 3, dx = tap % 3;
      const bf16_t* xj[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) xj[j] = Xp + swz((wm * 4 + j + dy) * PP_W + l16 + dx, g);
      mma_slab<XQ>(cur + wrow, xj, acc);
      if (next_slab) {
        RAW_BARRIER();                 // every wave is done with this slab's patch
        stash_x(slab + 1);
      }
      // the next tap's tile (issued one tap ago) must have landed; the one issued in this tap may stay in flight
      if (has2) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      RAW_BARRIER();
    }
  }

  const int ox = x0 + l16;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int oy = y0 + wm * 4 + j;
    if (oy >= p.H || ox >= p.W) continue;
    const int64_t m = ((int64_t)b * p.H + oy) * p.W + ox;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = nblk * SBN + wn * 64 + i * 16 + g * 4;
      if (n >= p.Cout) continue;
      store_out_quad(acc[i][j], p.bias ? p.bias + n : nullptr, p.res ? p.res + m * p.Cout + n : nullptr,
                     p.y + m * p.Cout + n, 0);
    }
  }
}

// packed fp32 weights [taps][Cin][cout_pad] -> per (tap, slab, cout block) tile: 3 planes x [128 cout][32 cin] bf16 in
// the swizzled LDS image.  One thread per 16-byte chunk of the output.
__global__ __launch_bounds__(256) void conv_split_weights_kernel(const float* __restrict__ wp, bf16_t* __restrict__ out,
                                                                 int taps, int Cin, int cout_pad) {
  const int kslabs = (Cin + SBK - 1) / SBK, nblks = cout_pad / SBN;
  const int64_t chunks = (int64_t)taps * kslabs * nblks * (TILE / 8);
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < chunks; idx += (int64_t)gridDim.x * blockDim.x) {
    const int within = (int)(idx % (TILE / 8));
    const int64_t tile = idx / (TILE / 8);
    const int plane = within / (PLANE / 8);
    const int rc = within % (PLANE / 8);
    const int r = rc >> 2, chunk = (rc & 3) ^ ((r >> 2) & 3);       // stored position -> logical chunk
    const int nb = (int)(tile % nblks);
    const int ks = (int)((tile / nblks) % kslabs);
    const int tap = (int)(tile / ((int64_t)nblks * kslabs));
    bf16_t o[8];
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
      const int c = ks * SBK + chunk * 8 + e;
      const float a = c < Cin ? wp[((int64_t)tap * Cin + c) * cout_pad + nb * SBN + r] : 0.f;
      const float b = c + 1 < Cin ? wp[((int64_t)tap * Cin + c + 1) * cout_pad + nb * SBN + r] : 0.f;
      uint32_t p1, p2, p3;
      split3_pair(a, b, p1, p2, p3);
      const uint32_t pk = plane == 0 ? p1 : plane == 1 ? p2 : p3;
      o[e] = (bf16_t)(pk & 0xffffu);
      o[e + 1] = (bf16_t)(pk >> 16);
    }
    *reinterpret_cast<uint4*>(out + idx * 8) = *reinterpret_cast<const uint4*>(o);
  }
}

// 64-channel workgroups when the 128-channel tiling would not give every CU two workgroups
void launch_split3(const SplitArgs& a, int64_t M, int64_t N, hipStream_t st) {
  const unsigned mt = (unsigned)(((M + SBM - 1) / SBM + 7) / 8 * 8);
  const int64_t wgs = ((M + SBM - 1) / SBM) * ((N + SBN - 1) / SBN);
  if (wgs < 512) hipLaunchKernelGGL(conv_split3_kernel<2>, dim3(mt, (unsigned)((N + 63) / 64)), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(conv_split3_kernel<4>, dim3(mt, (unsigned)((N + SBN - 1) / SBN)), dim3(256), 0, st, a);
}

}  // namespace

// ------------------------------------------------------------------------------------ C ABI
extern "C" int ug_conv_split_weights(const float* w_packed, uint16_t* w_split, int taps, int Cin, int cout_pad,
                                     hipStream_t st) {
  UG_REQUIRE(w_packed && w_split && taps > 0 && Cin > 0 && cout_pad > 0 && cout_pad % SBN == 0,
             "ug_conv_split_weights: needs cout_pad %% 128 == 0 (Cin=%d cout_pad=%d)", Cin, cout_pad);
  UG_REQUIRE(ug_aligned16(w_split), "ug_conv_split_weights: output must be 16-byte aligned");
  const int64_t chunks = (int64_t)taps * ((Cin + SBK - 1) / SBK) * (cout_pad / SBN) * (TILE / 8);
  int64_t g = (chunks + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(conv_split_weights_kernel, dim3((unsigned)g), dim3(256), 0, st, w_packed, (bf16_t*)w_split, taps, Cin,
                     cout_pad);
  UG_CHECK_LAUNCH("ug_conv_split_weights");
  return UG_OK;
}

extern "C" int ug_conv2d_split3(const float* x, const uint16_t* w_split, const float* bias, const float* residual, float* y,
                                int64_t B, int Hin, int Win, int Cin, int Cout, int cout_pad, int ksize, int stride,
                                int pad_top, int pad_left, int Hout, int Wout, int upsample2x, hipStream_t st) {
  UG_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && ksize >= 1 && ksize <= 16, "ug_conv2d_split3: bad shape");
  UG_REQUIRE(Cin % 4 == 0 && Cout % 4 == 0 && cout_pad % SBN == 0 && cout_pad >= Cout,
             "ug_conv2d_split3: needs Cin %% 4 == 0, Cout %% 4 == 0, cout_pad %% 128 == 0 (Cin=%d Cout=%d cout_pad=%d)", Cin,
             Cout, cout_pad);
  UG_REQUIRE(x && y && w_split && ug_aligned16(x) && ug_aligned16(y) && ug_aligned16(w_split) &&
                 (!bias || ug_aligned16(bias)) && (!residual || ug_aligned16(residual)),
             "ug_conv2d_split3: pointers must be 16-byte aligned");
  const int64_t M = B * Hout * Wout;
  UG_REQUIRE(M < (1LL << 31), "ug_conv2d_split3: too many output pixels");
  SplitArgs a{};
  a.x = x; a.w = (const bf16_t*)w_split; a.bias = bias; a.res = residual; a.y = y;
  a.B = (int)B; a.Hin = Hin; a.Win = Win; a.Cin = Cin; a.Hout = Hout; a.Wout = Wout; a.Cout = Cout;
  a.KH = ksize; a.KW = ksize; a.stride = stride; a.pad_t = pad_top; a.pad_l = pad_left; a.ups = upsample2x;
  a.nblks = cout_pad / SBN; a.M = (int)M; a.kslabs = (Cin + SBK - 1) / SBK;
  a.ldx = Cin; a.ldy = Cout; a.ldres = Cout;
  launch_split3(a, M, Cout, st);
  UG_CHECK_LAUNCH("ug_conv2d_split3");
  return UG_OK;
}

extern "C" int ug_linear_split3(const float* x, int64_t ldx, const uint16_t* w_split, const float* bias, const float* residual,
                                int64_t ldres, float* y, int64_t ldy, int64_t M, int64_t N, int64_t K, int n_pad, int act,
                                hipStream_t st) {
  UG_REQUIRE(M > 0 && N > 0 && K > 0 && (act == 0 || act == 1) && M < (1LL << 31), "ug_linear_split3: bad args");
  UG_REQUIRE(K % 4 == 0 && N % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && (!residual || ldres % 4 == 0) && n_pad % SBN == 0 &&
                 n_pad >= N,
             "ug_linear_split3: needs K, N and the row strides %% 4 == 0 and n_pad %% 128 == 0 (K=%lld N=%lld n_pad=%d)",
             (long long)K, (long long)N, n_pad);
  UG_REQUIRE(x && y && w_split && ug_aligned16(x) && ug_aligned16(y) && ug_aligned16(w_split) &&
                 (!bias || ug_aligned16(bias)) && (!residual || ug_aligned16(residual)),
             "ug_linear_split3: pointers must be 16-byte aligned");
  SplitArgs a{};
  a.x = x; a.w = (const bf16_t*)w_split; a.bias = bias; a.res = residual; a.y = y;
  a.B = (int)M; a.Hin = a.Win = a.Hout = a.Wout = 1; a.Cin = (int)K; a.Cout = (int)N;
  a.KH = a.KW = 1; a.stride = 1;
  a.nblks = n_pad / SBN; a.M = (int)M; a.kslabs = (int)((K + SBK - 1) / SBK);
  a.ldx = ldx; a.ldy = ldy; a.ldres = ldres; a.act = act;
  launch_split3(a, M, N, st);
  UG_CHECK_LAUNCH("ug_linear_split3");
  return UG_OK;
}

extern "C" int ug_conv3x3_split3(const float* x, const uint16_t* w_split, const float* bias, const float* residual, float* y,
                                 int64_t B, int H, int W, int Cin, int Cout, int cout_pad, const float* gn_mu_rstd,
                                 const float* gn_gamma, const float* gn_beta, int gn_groups, int gn_swish, hipStream_t st) {
  UG_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "ug_conv3x3_split3: bad shape");
  UG_REQUIRE(Cin % SBK == 0 && Cout % 4 == 0 && cout_pad % SBN == 0 && cout_pad >= Cout,
             "ug_conv3x3_split3: needs Cin %% 32 == 0, Cout %% 4 == 0, cout_pad %% 128 == 0 (Cin=%d Cout=%d cout_pad=%d)", Cin,
             Cout, cout_pad);
  UG_REQUIRE(x && y && w_split && ug_aligned16(x) && ug_aligned16(y) && ug_aligned16(w_split) &&
                 (!bias || ug_aligned16(bias)) && (!residual || ug_aligned16(residual)),
             "ug_conv3x3_split3: pointers must be 16-byte aligned");
  PatchArgs a{};
  a.x = x; a.w = (const bf16_t*)w_split; a.bias = bias; a.res = residual; a.y = y;
  a.B = (int)B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.nblks = cout_pad / SBN; a.kslabs = Cin / SBK;
  // 16-row tiles (eight waves, DMA-fed weights) when they still give every CU two workgroups' worth of work
  const int nb_n = (Cout + SBN - 1) / SBN;
  const int64_t big_tiles = B * ((W + PT_W - 1) / PT_W) * ((H + QT_H - 1) / QT_H);
  const bool big = big_tiles * nb_n >= 512;
  a.tiles_x = (W + PT_W - 1) / PT_W; a.tiles_y = big ? (H + QT_H - 1) / QT_H : (H + PT_H - 1) / PT_H;
  const int64_t ntiles = B * a.tiles_x * a.tiles_y;
  UG_REQUIRE(ntiles < (1LL << 30) && B * H * W < (1LL << 31), "ug_conv3x3_split3: too many output pixels");
  a.ntiles = (int)ntiles;
  // 64-channel workgroups when the 128-channel tiling of the 8-row variant would leave CUs without work
  const bool half = !big && ntiles * nb_n < 512;
  dim3 grid((unsigned)((ntiles + 7) / 8 * 8), (unsigned)(half ? (Cout + 63) / 64 : nb_n));
  const dim3 block(big ? 512 : 256);
  if (gn_mu_rstd) {
    UG_REQUIRE(gn_gamma && gn_beta && gn_groups > 0 && Cin % gn_groups == 0 && (Cin / gn_groups) % 4 == 0 &&
                   ug_aligned16(gn_gamma) && ug_aligned16(gn_beta) && ((uintptr_t)gn_mu_rstd & 7) == 0,
               "ug_conv3x3_split3: fused GroupNorm needs gamma/beta and channels-per-group %% 4 == 0 (Cin=%d groups=%d)", Cin,
               gn_groups);
    a.mu_rstd = reinterpret_cast<const float2*>(gn_mu_rstd); a.gamma = gn_gamma; a.beta = gn_beta;
    a.G = gn_groups; a.cpg = Cin / gn_groups; a.swish = gn_swish;
    if (big) hipLaunchKernelGGL(conv3x3_patch16_kernel<true>, grid, block, 0, st, a);
    else if (half) hipLaunchKernelGGL((conv3x3_patch_kernel<true, 2>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((conv3x3_patch_kernel<true, 4>), grid, block, 0, st, a);
  } else {
    if (big) hipLaunchKernelGGL(conv3x3_patch16_kernel<false>, grid, block, 0, st, a);
    else if (half) hipLaunchKernelGGL((conv3x3_patch_kernel<false, 2>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((conv3x3_patch_kernel<false, 4>), grid, block, 0, st, a);
  }
  UG_CHECK_LAUNCH("ug_conv3x3_split3");
  return UG_OK;
}
