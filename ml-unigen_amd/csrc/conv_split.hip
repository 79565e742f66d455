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
};

__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
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
  const int nblk = blockIdx.y;

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
  const int kslabs = p.Cin / SBK;
  const int nkt = p.KH * p.KW * kslabs;

  f32x4_t acc[4][4];                     // [n block][m block]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  float4 rx[4];
  bool kx[4];
  u32x4_t rw[6];
  int cur_tap = 0, cur_c0 = -SBK;
  const float* apix[4] = {p.x, p.x, p.x, p.x};
  bool aok[4] = {false, false, false, false};
  const u32x4_t* wt = reinterpret_cast<const u32x4_t*>(p.w) + (int64_t)nblk * (TILE / 8) + tid;
  const int64_t wt_step = (int64_t)p.nblks * (TILE / 8);

  auto fetch = [&]() __attribute__((always_inline)) {
    cur_c0 += SBK;
    if (cur_c0 >= p.Cin) { cur_c0 = 0; ++cur_tap; }
    if (cur_c0 == 0) {                   // new tap: im2col coordinates and bounds once per Cin/32 slabs
      const int dy = cur_tap / p.KW, dx = cur_tap % p.KW;
      const int He = p.ups ? p.Hin * 2 : p.Hin, We = p.ups ? p.Win * 2 : p.Win;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int iy = ay[i] * p.stride + dy - p.pad_t, ix = ax[i] * p.stride + dx - p.pad_l;
        aok[i] = av[i] && iy >= 0 && iy < He && ix >= 0 && ix < We;
        if (p.ups) { iy >>= 1; ix >>= 1; }
        apix[i] = aok[i] ? p.x + (((int64_t)ab[i] * p.Hin + iy) * p.Win + ix) * p.Cin : p.x;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      rx[i] = *reinterpret_cast<const float4*>(aok[i] ? apix[i] + cur_c0 + q * 4 : p.x);
      kx[i] = aok[i];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) rw[i] = wt[i * 256];
    wt += wt_step;
  };
  auto stash = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 6; ++i) reinterpret_cast<u32x4_t*>(Ws)[i * 256 + tid] = rw[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 v = kx[i] ? rx[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      uint32_t a1, a2, a3, b1, b2, b3;
      split3_pair(v.x, v.y, a1, a2, a3);
      split3_pair(v.z, v.w, b1, b2, b3);
      const int r = (tid >> 3) + i * 32;
      const int off = swz(r, q >> 1) + (q & 1) * 4;
      *reinterpret_cast<uint2*>(Xs + off) = make_uint2(a1, b1);
      *reinterpret_cast<uint2*>(Xs + PLANE + off) = make_uint2(a2, b2);
      *reinterpret_cast<uint2*>(Xs + 2 * PLANE + off) = make_uint2(a3, b3);
    }
  };

  fetch();
  stash();
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    if (kt + 1 < nkt) fetch();           // next slab's global loads stay in flight under the MFMAs
    const bf16_t* wl = Ws + swz(wn * 64 + l16, g);
    const bf16_t* xl = Xs + swz(wm * 64 + l16, g);
    bf16x8_t w1[4], w2[4], x1[4], x2[4], t[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w1[i] = *reinterpret_cast<const bf16x8_t*>(wl + i * 16 * SBK);
#pragma unroll
    for (int j = 0; j < 4; ++j) x1[j] = *reinterpret_cast<const bf16x8_t*>(xl + j * 16 * SBK);
    // smallest terms first
#pragma unroll
    for (int j = 0; j < 4; ++j) t[j] = *reinterpret_cast<const bf16x8_t*>(xl + 2 * PLANE + j * 16 * SBK);   // x3
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[i], t[j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i] = *reinterpret_cast<const bf16x8_t*>(wl + 2 * PLANE + i * 16 * SBK);   // w3
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(t[i], x1[j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) w2[i] = *reinterpret_cast<const bf16x8_t*>(wl + PLANE + i * 16 * SBK);
#pragma unroll
    for (int j = 0; j < 4; ++j) x2[j] = *reinterpret_cast<const bf16x8_t*>(xl + PLANE + j * 16 * SBK);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2[i], x2[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[i], x2[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2[i], x1[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[i], x1[j], acc[i][j], 0, 0, 0);
      }
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
    for (int i = 0; i < 4; ++i) {
      const int n = nblk * SBN + wn * 64 + i * 16 + g * 4;
      if (n >= p.Cout) continue;
      float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
      if (p.bias) {
        const float4 b = *reinterpret_cast<const float4*>(p.bias + n);
        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
      }
      if (p.res) {
        const float4 r = *reinterpret_cast<const float4*>(p.res + (int64_t)m * p.Cout + n);
        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
      }
      *reinterpret_cast<float4*>(p.y + (int64_t)m * p.Cout + n) = v;
    }
  }
}

// packed fp32 weights [taps][Cin][cout_pad] -> per (tap, slab, cout block) tile: 3 planes x [128 cout][32 cin] bf16 in
// the swizzled LDS image.  One thread per 16-byte chunk of the output.
__global__ __launch_bounds__(256) void conv_split_weights_kernel(const float* __restrict__ wp, bf16_t* __restrict__ out,
                                                                 int taps, int Cin, int cout_pad) {
  const int64_t chunks = (int64_t)taps * Cin * cout_pad * 3 / 8;
  const int kslabs = Cin / SBK, nblks = cout_pad / SBN;
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
      const float a = wp[((int64_t)tap * Cin + c) * cout_pad + nb * SBN + r];
      const float b = wp[((int64_t)tap * Cin + c + 1) * cout_pad + nb * SBN + r];
      uint32_t p1, p2, p3;
      split3_pair(a, b, p1, p2, p3);
      const uint32_t pk = plane == 0 ? p1 : plane == 1 ? p2 : p3;
      o[e] = (bf16_t)(pk & 0xffffu);
      o[e + 1] = (bf16_t)(pk >> 16);
    }
    *reinterpret_cast<uint4*>(out + idx * 8) = *reinterpret_cast<const uint4*>(o);
  }
}

}  // namespace

// ------------------------------------------------------------------------------------ C ABI
extern "C" int ug_conv_split_weights(const float* w_packed, uint16_t* w_split, int taps, int Cin, int cout_pad,
                                     hipStream_t st) {
  UG_REQUIRE(w_packed && w_split && taps > 0 && Cin > 0 && Cin % SBK == 0 && cout_pad > 0 && cout_pad % SBN == 0,
             "ug_conv_split_weights: needs Cin %% 32 == 0 and cout_pad %% 128 == 0 (Cin=%d cout_pad=%d)", Cin, cout_pad);
  UG_REQUIRE(ug_aligned16(w_split), "ug_conv_split_weights: output must be 16-byte aligned");
  const int64_t chunks = (int64_t)taps * Cin * cout_pad * 3 / 8;
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
  UG_REQUIRE(Cin % SBK == 0 && Cout % 4 == 0 && cout_pad % SBN == 0 && cout_pad >= Cout,
             "ug_conv2d_split3: needs Cin %% 32 == 0, Cout %% 4 == 0, cout_pad %% 128 == 0 (Cin=%d Cout=%d cout_pad=%d)", Cin,
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
  a.nblks = cout_pad / SBN; a.M = (int)M;
  dim3 grid((unsigned)(((M + SBM - 1) / SBM + 7) / 8 * 8), (unsigned)((Cout + SBN - 1) / SBN));
  hipLaunchKernelGGL(conv_split3_kernel, grid, dim3(256), 0, st, a);
  UG_CHECK_LAUNCH("ug_conv2d_split3");
  return UG_OK;
}
