// bf16 MFMA GEMM for gfx950:  C[M,N] = A[M,K] · B[N,K]^T   (both operands K-contiguous, "NT").
//
// This is the only dense-contraction kernel of the Qwen2.5 backbone (reference call sites:
// transformers Qwen2Attention q/k/v/o_proj, Qwen2MLP gate/up/down_proj and UniGen's lm_head,
// models/unigen.py:274-287).  Dgrad and wgrad reuse it: the host keeps W^T copies of the weights
// and transposes activations with ug_transpose_* so every contraction is K-contiguous.
//
// Structure: 128x128 tile, BK=64, 4 waves (2x2, 64x64 per wave = 4x4 mfma_f32_16x16x32_bf16
// fragments), operands staged HBM->LDS with 16-byte LDS-DMA (global_load_lds), two LDS buffers
// (tile k+1 in flight while tile k feeds the matrix pipe, one barrier per k-tile), XOR-swizzled
// LDS image (swizzle applied to the per-lane SOURCE address because LDS-DMA writes lane-linear),
// XCD-aware grouped tile order, operands passed swapped to the MFMA so each lane ends up with
// four consecutive output columns (8/16-byte epilogue stores).
#include "common.h"
#include "unigen_hip.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;   // 16 KiB per operand tile
constexpr int GROUP_M = 8;

enum Epi { EPI_BF16 = 0, EPI_F32 = 1, EPI_RESID = 2 };

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

struct GemmArgs {
  const bf16_t* A; const bf16_t* B; void* C;
  const bf16_t* bias;      // [N] bf16 or null          (EPI_BF16)
  const float* resid;      // [M,ldr] fp32              (EPI_RESID: C = resid + bf16round(acc))
  const float* alpha_dev;  // optional device scalar multiplying the accumulator (EPI_F32)
  int M, N, K;
  int64_t lda, ldb, ldc, ldr;
  int beta;                // EPI_F32: 1 => C += acc
  int tiles_m, tiles_n;
};

// LDS image of one operand tile: [128 rows][8 chunks of 16 B]; chunk c of row r lives at
// physical chunk c ^ ((r >> 1) & 7)  -> ds_read_b128 fragment reads are bank-conflict free.
__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

template <bool DMA>
__device__ __forceinline__ void stage_tile(const bf16_t* const (&src)[4], int koff, char* lds_tile,
                                           int wave, int lane) {
  // instruction i of wave w fills LDS bytes [(i*4+w)*1024, +1024): rows (i*4+w)*8 .. +8
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if constexpr (DMA) {
      char* dst = lds_tile + (i * 4 + wave) * 1024;   // wave-uniform; HW adds lane*16
      __builtin_amdgcn_global_load_lds((gptr_t)(src[i] + koff), (lptr_t)dst, 16, 0, 0);
    } else {
      const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(src[i] + koff);
      *reinterpret_cast<bf16x8_t*>(lds_tile + (i * 4 + wave) * 1024 + lane * 16) = v;
    }
  }
}

template <int EPI, bool DMA>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmArgs p) {
  __shared__ __attribute__((aligned(16))) char lds[4 * TILE_BYTES];   // A0 B0 A1 B1
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // ---- tile coordinates: XCD-contiguous chunks, GROUP_M row panels swept column-major
  const int nwg = p.tiles_m * p.tiles_n;
  const int pid = xcd_remap(blockIdx.x, nwg);
  const int per_group = GROUP_M * p.tiles_n;
  const int gid = pid / per_group, first_m = gid * GROUP_M;
  const int gsz = min(p.tiles_m - first_m, GROUP_M);
  const int tm = first_m + (pid % per_group) % gsz;
  const int tn = (pid % per_group) / gsz;
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- per-lane source pointers for the 4+4 staging instructions (rows clamped: an out-of-range
  // row re-reads the last valid row, its products are discarded by the guarded epilogue)
  const bf16_t* srcA[4]; const bf16_t* srcB[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (i * 4 + wave) * 8 + (lane >> 3);
    const int chunk = swz(row, lane & 7);           // logical k-chunk this lane fetches
    const int ra = min(m0 + row, p.M - 1), rb = min(n0 + row, p.N - 1);
    srcA[i] = p.A + (int64_t)ra * p.lda + chunk * 8;
    srcB[i] = p.B + (int64_t)rb * p.ldb + chunk * 8;
  }

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // fragment read offsets (bytes inside a tile) for k-step 0; k-step 1 flips chunk bit 2
  int offA[4], offB[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ra = wm * 64 + i * 16 + (lane & 15), rb = wn * 64 + i * 16 + (lane & 15);
    offA[i] = ra * 128 + swz(ra, lane >> 4) * 16;
    offB[i] = rb * 128 + swz(rb, lane >> 4) * 16;
  }

  const int nk = p.K / BK;
  stage_tile<DMA>(srcA, 0, lds, wave, lane);
  stage_tile<DMA>(srcB, 0, lds + TILE_BYTES, wave, lane);
  __syncthreads();   // hipcc drains vmcnt(0) for in-flight LDS-DMA here

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
      char* nxt = lds + (cur ^ 1) * 2 * TILE_BYTES;
      stage_tile<DMA>(srcA, (kt + 1) * BK, nxt, wave, lane);
      stage_tile<DMA>(srcB, (kt + 1) * BK, nxt + TILE_BYTES, wave, lane);
    }
    const char* tA = lds + cur * 2 * TILE_BYTES;
    const char* tB = tA + TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fa[i] = *reinterpret_cast<const bf16x8_t*>(tA + (offA[i] ^ (ks << 6)));
        fb[i] = *reinterpret_cast<const bf16x8_t*>(tB + (offB[i] ^ (ks << 6)));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)   // swapped operands: lane gets C[m=lane&15][n=(lane>>4)*4+r]
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

  // ---- epilogue
  float alpha = 1.f;
  if constexpr (EPI == EPI_F32) { if (p.alpha_dev) alpha = *p.alpha_dev; }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + (lane & 15);
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
      if (n >= p.N) continue;
      f32x4_t v = acc[i][j];
      const bool full = (n + 3 < p.N);
      if constexpr (EPI == EPI_BF16) {
        bf16_t* c = reinterpret_cast<bf16_t*>(p.C) + (int64_t)m * p.ldc + n;
        if (p.bias) {
#pragma unroll
          for (int r = 0; r < 4; ++r) if (n + r < p.N) v[r] += bf2f(p.bias[n + r]);
        }
        if (full) {
          uint2 o; o.x = pack_bf2(v[0], v[1]); o.y = pack_bf2(v[2], v[3]);
          *reinterpret_cast<uint2*>(c) = o;
        } else {
          for (int r = 0; r < 4; ++r) if (n + r < p.N) c[r] = f2bf(v[r]);
        }
      } else if constexpr (EPI == EPI_F32) {
        float* c = reinterpret_cast<float*>(p.C) + (int64_t)m * p.ldc + n;
        if (full) {
          float4 o = make_float4(v[0] * alpha, v[1] * alpha, v[2] * alpha, v[3] * alpha);
          if (p.beta) { const float4 old = *reinterpret_cast<const float4*>(c);
                        o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w; }
          *reinterpret_cast<float4*>(c) = o;
        } else {
          for (int r = 0; r < 4; ++r) if (n + r < p.N) c[r] = (p.beta ? c[r] : 0.f) + v[r] * alpha;
        }
      } else {  // EPI_RESID: fp32 residual stream += bf16-rounded projection (reference: bf16 Linear
                // output added to the fp32 residual under autocast)
        float* c = reinterpret_cast<float*>(p.C) + (int64_t)m * p.ldc + n;
        const float* rs = p.resid + (int64_t)m * p.ldr + n;
        if (full) {
          const float4 old = *reinterpret_cast<const float4*>(rs);
          float4 o;
          o.x = old.x + bf2f(f2bf(v[0])); o.y = old.y + bf2f(f2bf(v[1]));
          o.z = old.z + bf2f(f2bf(v[2])); o.w = old.w + bf2f(f2bf(v[3]));
          *reinterpret_cast<float4*>(c) = o;
        } else {
          for (int r = 0; r < 4; ++r) if (n + r < p.N) c[r] = rs[r] + bf2f(f2bf(v[r]));
        }
      }
    }
  }
}

template <int EPI>
int launch(const GemmArgs& a, int variant, hipStream_t st) {
  dim3 grid(a.tiles_m * a.tiles_n), block(256);
  if (variant == 1) hipLaunchKernelGGL((gemm_nt_kernel<EPI, false>), grid, block, 0, st, a);
  else hipLaunchKernelGGL((gemm_nt_kernel<EPI, true>), grid, block, 0, st, a);
  UG_CHECK_LAUNCH("ug_gemm_bf16_nt");
  return UG_OK;
}

int g_gemm_variant = 0;   // 0 = LDS-DMA staging (default), 1 = register staging (validation arm)

}  // namespace

extern "C" int ug_gemm_set_variant(int v) { g_gemm_variant = v; return UG_OK; }

extern "C" int ug_gemm_bf16_nt(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                               int64_t M, int64_t N, int64_t K, int epilogue, const void* bias,
                               const float* resid, int64_t ldr, int beta, const float* alpha_dev,
                               hipStream_t stream) {
  UG_REQUIRE(M > 0 && N > 0 && K > 0, "ug_gemm_bf16_nt: empty problem M=%ld N=%ld K=%ld", (long)M, (long)N, (long)K);
  UG_REQUIRE(K % BK == 0, "ug_gemm_bf16_nt: K=%ld must be a multiple of %d (pad the contraction dim)", (long)K, BK);
  UG_REQUIRE(lda % 8 == 0 && ldb % 8 == 0, "ug_gemm_bf16_nt: lda/ldb must be multiples of 8 elements");
  UG_REQUIRE(ug_aligned16(A) && ug_aligned16(B) && ug_aligned16(C), "ug_gemm_bf16_nt: A/B/C must be 16-byte aligned");
  UG_REQUIRE(ldc % 4 == 0, "ug_gemm_bf16_nt: ldc must be a multiple of 4 elements");
  UG_REQUIRE(M < (1 << 30) && N < (1 << 30), "ug_gemm_bf16_nt: dims too large");
  GemmArgs a;
  a.A = (const bf16_t*)A; a.B = (const bf16_t*)B; a.C = C;
  a.bias = (const bf16_t*)bias; a.resid = resid; a.alpha_dev = alpha_dev;
  a.M = (int)M; a.N = (int)N; a.K = (int)K;
  a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.ldr = ldr; a.beta = beta;
  a.tiles_m = (int)((M + BM - 1) / BM); a.tiles_n = (int)((N + BN - 1) / BN);
  switch (epilogue) {
    case EPI_BF16: return launch<EPI_BF16>(a, g_gemm_variant, stream);
    case EPI_F32: return launch<EPI_F32>(a, g_gemm_variant, stream);
    case EPI_RESID:
      UG_REQUIRE(resid != nullptr && ldr % 4 == 0, "ug_gemm_bf16_nt: EPI_RESID needs a 16B-aligned residual");
      return launch<EPI_RESID>(a, g_gemm_variant, stream);
    default: ug_set_error("ug_gemm_bf16_nt: unknown epilogue %d", epilogue); return UG_ERR_ARG;
  }
}
