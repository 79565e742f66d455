// bf16 MFMA GEMM for gfx950, fp32 accumulate:   C[M,N] = opA . opB^T
//
// The only dense-contraction kernel of the Qwen2.5 backbone (reference call sites: transformers
// Qwen2Attention q/k/v/o_proj, Qwen2MLP gate/up/down_proj, UniGen's tied lm_head --
// models/unigen.py:274-287 -- and their autograd dgrad / wgrad).  Each operand can be given in
// either storage order, so forward, dgrad and wgrad all run on it with no transposed copies:
//     row-major "RowK":  X[row][k]   (k contiguous)          A: [M][K] lda      B: [N][K] ldb
//     k-major   "KRow":  X[k][row]   (row contiguous)        A: [K][M] lda      B: [K][N] ldb
//   forward  y = x W^T        : A = x   RowK, B = W  RowK
//   dgrad    dx = dy W        : A = dy  RowK, B = W  KRow   (W is [N_out][K_in] = [k][row])
//   wgrad    dW = dy^T x      : A = dy  KRow, B = x  KRow   (contraction over tokens)
//
// Kernels in this file (the launcher picks per shape, see launch<>() and DESIGN.md section 4.1):
//   gemm_kernel             128x128 tiles, 4 waves, 1-2 LDS stages: small / ragged problems
//   gemm_kernel_p8          256x256 tiles, 8 waves in two groups that alternate fragment reads (L) and MFMAs (M), 4-stage
//                           LDS-DMA ring; two barriers per k-tile (forward, dgrad) or one (weight gradients); k-sliced forms
//                           for partial last rounds and small outputs
//   gemm_kernel_p8_wgrad_group   the same loop over the tiles of up to 8 weight-gradient problems sharing K (one launch
//                           per decoder layer)
//   gemm_kernel_p10         192 ... 320 x 256 tiles (A row-major): outputs whose 256-row tiling leaves a mostly empty round; the
//                           tile height is the smallest that keeps a one-round launch in one round
//
// The 128x128 kernel: BK=64, 4 waves (2x2, 64x64 per wave = 4x4 mfma_f32_16x16x32_bf16
// fragments), operands staged HBM->LDS with 16-byte LDS-DMA (global_load_lds).  Two pipelining
// modes, chosen per launch (measured on MI355X, tools/gemm_bench.py):
//   single LDS stage (32 KiB, <=128 VGPRs -> 4 workgroups/CU): latency is hidden by the other resident
//     workgroups; fastest for almost every backbone shape (e.g. dgrad 1536x8960: 1017 vs 772 TF/s);
//   two LDS stages (64 KiB, 2 workgroups/CU, tile k+1's DMA under tile k's MFMAs): kept for long-K
//     row-major forward GEMMs where it is ~5 % ahead.
// A 256x128 / 8-wave / 3-stage counted-vmcnt variant was measured 10-25 % SLOWER on every shape
// (one lock-stepped workgroup per CU) and removed.  Small weight-gradient outputs (fewer tiles than
// CUs) are split along K across gridDim.y with fp32 atomics into the (already accumulating) output.
// LDS images are
// XOR-swizzled through the per-lane SOURCE address (LDS-DMA writes lane-linear):
//     RowK tile [128 rows][8 chunks]:  chunk ^= (row>>1)&7        -> conflict-free ds_read_b128
//     KRow tile [64 k][16 chunks]:     chunk ^= h(k)<<1, h(k) = (k&3)|((k>>3)&1)<<2
//                                      -> conflict-free ds_read_b64_tr_b16 (hardware transpose read:
//                                         lane i of a 16-lane group receives column i of the 4x16
//                                         block whose row j is the 4 pieces addressed by lanes 4j..4j+3)
// Out-of-range contraction rows/chunks are fetched from a zero page, so K needs no padding in KRow
// mode; M/N edges are clamped on load and guarded on store.  XCD-aware grouped tile order; operands
// are passed swapped to the MFMA so each lane ends up with four consecutive output columns.
#include "common.h"
#include <type_traits>
#include <atomic>
#include "unigen_hip.h"
#include "vmem_asm.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;   // 16 KiB per operand tile (either layout)
constexpr int GROUP_M = 8;
constexpr int P8_GROUP_M = 4;    // row panels per column sweep of the 256-wide kernels: the four 786 KB A panels of a
                                             // K = 1536 launch stay in an XCD's 4 MB L2 while its 32 workgroups walk the columns
                                             // (8: 8192^3 1325 -> 1385, gate_up forward / weight gradients +1..2 %)

enum Epi { EPI_BF16 = 0, EPI_F32 = 1, EPI_RESID = 2, EPI_ROPE = 3, EPI_SWIGLU = 4, EPI_SWIGLU_BWD = 5 };    // EPI_ROPE: EPI_BF16 + rotate-half RoPE on the first rope_cols columns (128...320-row kernel only)

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;

__device__ __attribute__((aligned(16))) uint32_t g_zero_page[64];   // 256 B of zeros: source of masked DMA lanes

struct GemmArgs {
  const bf16_t* A; const bf16_t* B; void* C;
  const bf16_t* bias;      // [N] bf16 or null          (EPI_BF16)
  const float* resid;      // [M,ldr] fp32              (EPI_RESID: C = resid + bf16round(acc))
  const float* alpha_dev;  // optional device scalar multiplying the accumulator (EPI_F32)
  int M, N, K;
  int64_t lda, ldb, ldc, ldr;
  int beta;                // EPI_F32: 1 => C += acc
  const bf16_t* sw_gu; int64_t ld_gu;      // EPI_SWIGLU_BWD: the forward's gate | up [M, 2 * swiglu_I] (C = d(gate | up))
  const float* rope_cos; const float* rope_sin; int rope_L, rope_cols;     // EPI_ROPE: tables [rope_L, 64] fp32, row m sits at position m % rope_L
  int tiles_m, tiles_n;
  // staggered kernel, partial last round: the first `full_tiles` tiles run whole, every remaining tile is cut into
  // `tail_split` k-slices that add into the fp32 scratch tail_ws[tile - full_tiles][256][256] (finished by tail_finish)
  int full_tiles, tail_split;
  float* tail_ws;
  int tail_private;        // 1: every k-slice stores its own [256][256] partial (no atomics); the finisher sums them
  int one_barrier;         // 256x256 kernel: one barrier per k-tile (A/B runs; policy bit 0x200)
  int wide_epilogue;       // 256x256 kernel: LDS-transposed 16-byte stores (0 only for A/B runs, ug_gemm_set_tile_policy(100))
  // fused SwiGLU (ug_gemm_bf16_swiglu): B = [gate rows | up rows] of the fused weight, I rows each; a tile's 256 columns are
  // 128 gate columns and the SAME 128 hidden units' up columns; C = gu [M][2I], act [M][I] = bf16(bf16(silu(gate)) * up)
  int swiglu_I;
  bf16_t* act; int64_t ld_act;
};

__device__ __forceinline__ int swz_rowk(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }
__device__ __forceinline__ int swz_krow(int k, int chunk) { return chunk ^ ((((k & 3) | (((k >> 3) & 1) << 2))) << 1); }

// Per-lane staging plan for one operand tile of ROWS rows x 64 k, staged by NW waves: NI = ROWS/(8*NW)
// 1-KiB DMA instructions per wave per k-tile.  Lanes whose source lies outside the operand point at
// the zero page with a zero stride, so the steady-state loop is branch-free; only a ragged LAST k-tile
// takes the checked path.
template <bool KMAJOR, int ROWS, int NW>
struct Stager {
  static constexpr int NI = ROWS / (8 * NW);
  const bf16_t* src[NI];   // source for k-tile 0 (zero page for statically masked lanes)
  int64_t step[NI];        // elements to advance per k-tile (0 for zero-page lanes)
  int kofs[NI];            // contraction offset inside the tile of what this lane fetches (k-row, or chunk*8)
  __device__ __forceinline__ void init(const bf16_t* X, int64_t ld, int row0, int rows_total, int wave, int lane) {
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page);
    constexpr int CPR = ROWS / 8;          // 16-byte chunks per k-row of a k-major tile
    constexpr int KPI = 64 / CPR;          // k-rows covered by one 1-KiB instruction
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int inst = i * NW + wave;
      if constexpr (!KMAJOR) {
        const int row = inst * 8 + (lane >> 3);
        const int chunk = swz_rowk(row, lane & 7);
        const int r = min(row0 + row, rows_total - 1);          // clamped: products land in discarded rows
        src[i] = X + (int64_t)r * ld + chunk * 8;
        step[i] = BK;
        kofs[i] = chunk * 8;
      } else {
        const int k = inst * KPI + lane / CPR;
        const int chunk = swz_krow(k, lane % CPR);
        const int col = row0 + chunk * 8;
        const bool ok = (col + 8 <= ld);
        src[i] = ok ? X + (int64_t)k * ld + col : zero;
        step[i] = ok ? (int64_t)BK * ld : 0;
        kofs[i] = k;
      }
    }
  }
  template <bool CHECK>
  __device__ __forceinline__ void issue(int kt, int K, char* lds_tile, int wave) const {
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const bf16_t* s = src[i] + kt * step[i];
      if constexpr (CHECK) {
        // past the contraction extent: zero page.  (row-major operands are fetched in 8-element chunks;
        // the chunk straddling K is read whole -- its tail must be finite, see the entry-point contract)
        const bool past = kt * BK + kofs[i] >= (KMAJOR ? K : ((K + 7) & ~7));
        s = reinterpret_cast<const bf16_t*>(past ? reinterpret_cast<uintptr_t>(zero) : reinterpret_cast<uintptr_t>(s));
      }
      char* dst = lds_tile + (i * NW + wave) * 1024;            // wave-uniform; HW adds lane*16
      __builtin_amdgcn_global_load_lds((gptr_t)s, (lptr_t)dst, 16, 0, 0);
    }
  }
};

// MFMA fragment (16 rows x 32 k) for rows r0.., k-step ks, from either LDS image (ROWS = tile rows)
template <bool KMAJOR, int ROWS>
__device__ __forceinline__ bf16x8_t load_frag(const char* tile, int r0, int ks, int lane) {
  if constexpr (!KMAJOR) {
    const int row = r0 + (lane & 15);
    const int chunk = swz_rowk(row, ks * 4 + (lane >> 4));
    return *reinterpret_cast<const bf16x8_t*>(tile + row * 128 + chunk * 16);
  } else {
    const int i16 = lane & 15, g = lane >> 4;
    const int k = ks * 32 + g * 8 + (i16 >> 2);                 // rows k..k+3 <-> lanes 4j..4j+3 of the group
    const int chunk = (r0 >> 3) + ((i16 & 3) >> 1);
    const int sub = (i16 & 1) * 8;
    const char* p0 = tile + k * (ROWS * 2) + swz_krow(k, chunk) * 16 + sub;
    const char* p1 = tile + (k + 4) * (ROWS * 2) + swz_krow(k + 4, chunk) * 16 + sub;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p0);
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p1);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  }
}

// Epilogue for one wave's 64x64 block of accumulators (swapped-operand C layout: lane holds
// C[m = lane&15 (+16i)][n = (lane>>4)*4 + r (+16j)]).
template <int EPI>
__device__ __forceinline__ void store_tile(const GemmArgs& p, f32x4_t (&acc)[4][4], int mbase, int nbase, int lane,
                                           bool split) {
  float alpha = 1.f;
  if constexpr (EPI == EPI_F32) { if (p.alpha_dev) alpha = *p.alpha_dev; }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = mbase + i * 16 + (lane & 15);
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = nbase + j * 16 + (lane >> 4) * 4;
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
        if (split) {         // split-K partial: accumulate with device-scope fp32 atomics (beta == 1 by contract)
          for (int r = 0; r < 4; ++r) if (n + r < p.N) atomicAdd(c + r, v[r] * alpha);
        } else if (full) {
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

// Epilogue of the 256x256 kernel for interior column panels: each wave turns its 128x64 accumulator block through a private
// LDS strip (the operand ring is idle by then) 32 rows at a time, so that every global store is 16 bytes per lane and a
// wave writes whole 128-byte (bf16) / 256-byte (fp32) row segments -- 16 dwordx4 stores per wave instead of 32 dwordx2
// to 32-byte pieces of 16 different rows each (the no-store ablation put the scattered form at ~20 % of a K = 1536 GEMM).
//   strip pitch: bf16 144 B (36 dwords: the 16 rows of a ds_write_b64 group fall on 8 bank quads, 2-way at worst; rows stay
//   16-byte aligned for the ds_read_b128 that follows), fp32 272 B (68 dwords: ds_write_b128's 8-lane groups conflict-free).
// 16-byte epilogue store (a non-temporal form measured neutral in the step: tools/probes/probe_switches.patch, -DUG_EPI_NT)
#define UG_ST16(ptr, val) (*reinterpret_cast<uint4*>(ptr) = (val))
typedef unsigned int gnt4_t __attribute__((ext_vector_type(4)));
constexpr int EP_ROWS = 32;
constexpr int EP_PITCH_BF16 = 144, EP_PITCH_F32 = 272;
constexpr int EP_STRIP = EP_ROWS * EP_PITCH_F32;            // 8704 B per wave (the bf16 strip needs 4608)

// One rotate-half pair exactly as rope_kernel (elementwise.hip) computes it: products and sums round separately, contraction off.
__device__ __forceinline__ void rope_pair(float x1, float x2, float c, float s, float& o1, float& o2) {
#pragma clang fp contract(off)
  const float a1 = x1 * c, a2 = x2 * c;
  const float b1 = x2 * s, b2 = x1 * s;
  o1 = a1 - b1; o2 = a2 + b2;
}

// EPI_ROPE (fused q/k/v projection, round 4): the wave's 64 columns are TWO 32-column pieces of one 128-wide head, 64 apart
// (nbase .. nbase + 31 and nbase + 64 .. nbase + 95; accumulator column blocks 0, 1 | 2, 3), so the partner x[j + 64] of a
// rotate-half pair is the same lane's element of block j + 2: RoPE is register arithmetic between the bf16 rounding of
// (acc + bias) and the bf16 rounding of the result -- the same two roundings as ug_gemm_bf16 followed by ug_rope, bit for bit.
// rope_p0 = column of the wave's first piece inside the head's first half (0 or 32); roped = this tile's columns are q / k heads.
template <int EPI, int MF = 8>                              // MF = 16-row accumulator blocks of the wave (8, or 10 for 320-row tiles)
__device__ __forceinline__ void store_tile_lds(const GemmArgs& p, f32x4_t (&acc)[MF][4], char* strip, int mbase, int nbase, int lane,
                                               int rope_p0 = 0, bool roped = false) {
  constexpr bool BF = EPI == EPI_BF16 || EPI == EPI_ROPE;
  float alpha = 1.f;
  if constexpr (EPI == EPI_F32) { if (p.alpha_dev) alpha = *p.alpha_dev; }
  float bias4[4][4];
  if constexpr (BF) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint2 b = make_uint2(0u, 0u);                          // 4 bf16 of this lane's columns (8-byte aligned: the flat
      const int cj = EPI == EPI_ROPE ? (j & 1) * 16 + (j >> 1) * 64 : j * 16;
      if (p.bias) b = *reinterpret_cast<const uint2*>(p.bias + nbase + cj + (lane >> 4) * 4);    // buffers are 64-element aligned)
      bias4[j][0] = __uint_as_float(b.x << 16); bias4[j][1] = __uint_as_float(b.x & 0xffff0000u);
      bias4[j][2] = __uint_as_float(b.y << 16); bias4[j][3] = __uint_as_float(b.y & 0xffff0000u);
    }
  }
#pragma unroll
  for (int c = 0; c < (MF + 1) / 2; ++c) {                   // 32-row chunk = accumulator row blocks 2c, 2c+1 (an odd MF ends on a half chunk)
    const int rows_here = (2 * c + 1 < MF) ? 32 : 16;
    // ---- registers -> strip (MFMA layout: lane holds rows lane&15, 4 consecutive columns per block)
#pragma unroll
    for (int ii = 0; ii < 2; ++ii) {
      if (2 * c + ii >= MF) break;
      const int row = ii * 16 + (lane & 15);
      if constexpr (EPI == EPI_ROPE) {
        if (roped) {
          const int pos = min(mbase + c * EP_ROWS + row, p.M - 1) % p.rope_L;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int pc = rope_p0 + j * 16 + (lane >> 4) * 4;
            const float4 cs = *reinterpret_cast<const float4*>(p.rope_cos + (int64_t)pos * 64 + pc);
            const float4 sn = *reinterpret_cast<const float4*>(p.rope_sin + (int64_t)pos * 64 + pc);
            const float cc[4] = {cs.x, cs.y, cs.z, cs.w}, ss[4] = {sn.x, sn.y, sn.z, sn.w};
            f32x4_t& lo = acc[2 * c + ii][j];
            f32x4_t& hi = acc[2 * c + ii][j + 2];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const float x1 = bf2f(f2bf(lo[k] + bias4[j][k])), x2 = bf2f(f2bf(hi[k] + bias4[j + 2][k]));
              float o1, o2;
              rope_pair(x1, x2, cc[k], ss[k], o1, o2);
              lo[k] = o1; hi[k] = o2;                              // (the bias is already inside x1 / x2)
            }
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = j * 16 + (lane >> 4) * 4;
        const f32x4_t v = acc[2 * c + ii][j];
        if constexpr (EPI == EPI_ROPE) {
          uint2 o;
          if (roped) { o.x = pack_bf2(v[0], v[1]); o.y = pack_bf2(v[2], v[3]); }
          else { o.x = pack_bf2(v[0] + bias4[j][0], v[1] + bias4[j][1]); o.y = pack_bf2(v[2] + bias4[j][2], v[3] + bias4[j][3]); }
          *reinterpret_cast<uint2*>(strip + row * EP_PITCH_BF16 + col * 2) = o;
        } else if constexpr (EPI == EPI_BF16) {
          uint2 o;
          o.x = pack_bf2(v[0] + bias4[j][0], v[1] + bias4[j][1]);
          o.y = pack_bf2(v[2] + bias4[j][2], v[3] + bias4[j][3]);
          *reinterpret_cast<uint2*>(strip + row * EP_PITCH_BF16 + col * 2) = o;
        } else {
          *reinterpret_cast<float4*>(strip + row * EP_PITCH_F32 + col * 4) = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the strip is wave-private: no barrier, only this wave's writes
    // ---- strip -> global, 16 bytes per lane, whole row segments
    if constexpr (BF) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = q * 8 + (lane >> 3), ch = lane & 7;
        const int m = mbase + c * EP_ROWS + row;
        const uint4 v = *reinterpret_cast<const uint4*>(strip + row * EP_PITCH_BF16 + ch * 16);
        const int gc = EPI == EPI_ROPE ? (ch & 3) * 8 + (ch >> 2) * 64 : ch * 8;        // (two 64-byte pieces per row with EPI_ROPE)
        if (m < p.M && row < rows_here) UG_ST16(reinterpret_cast<bf16_t*>(p.C) + (int64_t)m * p.ldc + nbase + gc, v);
      }
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int row = q * 4 + (lane >> 4), ch = lane & 15;
        const int m = mbase + c * EP_ROWS + row;
        float4 v = *reinterpret_cast<const float4*>(strip + row * EP_PITCH_F32 + ch * 16);
        if (m < p.M && row < rows_here) {
          float* cptr = reinterpret_cast<float*>(p.C) + (int64_t)m * p.ldc + nbase + ch * 4;
          if constexpr (EPI == EPI_F32) {
            v.x *= alpha; v.y *= alpha; v.z *= alpha; v.w *= alpha;
            if (p.beta) { const float4 old = *reinterpret_cast<const float4*>(cptr); v.x += old.x; v.y += old.y; v.z += old.z; v.w += old.w; }
          } else {
            const float4 old = *reinterpret_cast<const float4*>(p.resid + (int64_t)m * p.ldr + nbase + ch * 4);
            v.x = old.x + bf2f(f2bf(v.x)); v.y = old.y + bf2f(f2bf(v.y)); v.z = old.z + bf2f(f2bf(v.z)); v.w = old.w + bf2f(f2bf(v.w));
          }
          *reinterpret_cast<float4*>(cptr) = v;
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // strip reads retired before the next chunk overwrites it
  }
}

__device__ __forceinline__ float silu_gemm(float g) { return silu_train(g); }

// SwiGLU in the epilogue of the 128 ... 320-row kernel (EPI_SWIGLU, round 4) WITHOUT any exchange between waves: the B tile's rows
// 0..127 are 128 gate units and rows 128..255 the up projection of the SAME hidden units (Stager32's split_rows), and a wave's four
// accumulator column blocks are gate units hbase .. hbase + 31 (blocks 0, 1) and their up partners (blocks 2, 3) -- gate and up of
// a hidden unit sit in the same lane two blocks apart.  act = bf16(bf16(silu(bf16 gate)) * bf16 up) as swiglu_fwd_kernel computes
// it; gu leaves through the wave's strip as two 64-byte pieces per row (gate | up), act as one.
template <int MF>
__device__ __forceinline__ void store_tile_swiglu_perm(const GemmArgs& p, f32x4_t (&acc)[MF][4], char* strip, int mbase, int hbase, int lane) {
  constexpr int ACT_PITCH = 80;                               // 32 bf16 + 16 B: rows stay 16-byte aligned
  char* astrip = strip + EP_ROWS * EP_PITCH_BF16;             // 4608 + 32 x 80 = 7168 B <= EP_STRIP
  bf16_t* C = reinterpret_cast<bf16_t*>(p.C);
#pragma unroll
  for (int c = 0; c < (MF + 1) / 2; ++c) {
    const int rows_here = (2 * c + 1 < MF) ? 32 : 16;
#pragma unroll
    for (int ii = 0; ii < 2; ++ii) {
      if (2 * c + ii >= MF) break;
      const int row = ii * 16 + (lane & 15);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const f32x4_t gv = acc[2 * c + ii][j], uv = acc[2 * c + ii][j + 2];
        float a[4];
        uint32_t gb[4], ub[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          gb[k] = f2bf(gv[k]); ub[k] = f2bf(uv[k]);
          const float s = bf2f(f2bf(silu_gemm(bf2f((bf16_t)gb[k]))));
          a[k] = s * bf2f((bf16_t)ub[k]);
        }
        const int col = j * 16 + (lane >> 4) * 4;
        *reinterpret_cast<uint2*>(strip + row * EP_PITCH_BF16 + col * 2) = make_uint2(gb[0] | (gb[1] << 16), gb[2] | (gb[3] << 16));
        *reinterpret_cast<uint2*>(strip + row * EP_PITCH_BF16 + (32 + col) * 2) = make_uint2(ub[0] | (ub[1] << 16), ub[2] | (ub[3] << 16));
        *reinterpret_cast<uint2*>(astrip + row * ACT_PITCH + col * 2) = make_uint2(pack_bf2(a[0], a[1]), pack_bf2(a[2], a[3]));
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int q = 0; q < 4; ++q) {                             // gu: 32 rows x (4 gate + 4 up) 16-byte pieces
      const int row = q * 8 + (lane >> 3), ch = lane & 7;
      const int m = mbase + c * EP_ROWS + row;
      const uint4 v = *reinterpret_cast<const uint4*>(strip + row * EP_PITCH_BF16 + ch * 16);
      const int gc = (ch >> 2) * p.swiglu_I + hbase + (ch & 3) * 8;
      if (m < p.M && row < rows_here) UG_ST16(C + (int64_t)m * p.ldc + gc, v);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {                             // act: 32 rows x 4 pieces
      const int row = q * 16 + (lane >> 2), ch = lane & 3;
      const int m = mbase + c * EP_ROWS + row;
      const uint4 v = *reinterpret_cast<const uint4*>(astrip + row * ACT_PITCH + ch * 16);
      if (m < p.M && row < rows_here)
        UG_ST16(p.act + (int64_t)m * p.ld_act + hbase + ch * 8, v);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

// SwiGLU backward in the epilogue of the down projection's dgrad (EPI_SWIGLU_BWD, round 4): the tile is d(act) for 256 hidden units;
// it is rounded to bf16 (what the two-launch form stores), turned through the wave's strip into 16-byte row pieces, and each lane
// combines its 8 values with the forward's gate and up values of the same units (two 16-byte loads) into d(gate) and d(up) -- two
// 16-byte stores into d(gate | up).  d(act) itself never reaches HBM and the swiglu_bwd pass (1.1 GB per layer) disappears.
// The gate / up loads of chunk c + 1 are issued before chunk c is processed; they are inline assembly with hand-counted waits
// (loads and stores in flight together make hipcc drain vmcnt(0), see adamw_lean2_kernel): at most the 8 loads of the next chunk
// may be outstanding when a chunk's operands are used, which also retires the previous chunk's stores.
typedef unsigned int gu4_t __attribute__((ext_vector_type(4)));
template <int MF>
__device__ __forceinline__ void store_tile_swiglu_bwd(const GemmArgs& p, f32x4_t (&acc)[MF][4], char* strip, int mbase, int nbase, int lane) {
  constexpr int NCH = (MF + 1) / 2;
  bf16_t* C = reinterpret_cast<bf16_t*>(p.C);
  const int I = p.swiglu_I;
  gu4_t gq[2][4], uq[2][4];
  auto fetch = [&](int c, gu4_t (&g)[4], gu4_t (&u)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = q * 8 + (lane >> 3), ch = lane & 7;
      const int m = min(mbase + c * EP_ROWS + row, p.M - 1);
      const bf16_t* gp = p.sw_gu + (int64_t)m * p.ld_gu + nbase + ch * 8;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(g[q]) : "v"(gp) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(u[q]) : "v"(gp + I) : "memory");
    }
  };
  fetch(0, gq[0], uq[0]);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int rows_here = (2 * c + 1 < MF) ? 32 : 16;
    gu4_t (&g)[4] = gq[c & 1];
    gu4_t (&u)[4] = uq[c & 1];
    if (c + 1 < NCH) fetch(c + 1, gq[(c + 1) & 1], uq[(c + 1) & 1]);
#pragma unroll
    for (int ii = 0; ii < 2; ++ii) {
      if (2 * c + ii >= MF) break;
      const int row = ii * 16 + (lane & 15);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4_t v = acc[2 * c + ii][j];
        *reinterpret_cast<uint2*>(strip + row * EP_PITCH_BF16 + (j * 16 + (lane >> 4) * 4) * 2) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
      }
    }
    if (c + 1 < NCH)
      asm volatile("s_waitcnt vmcnt(8)" : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]), "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]) : : "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]), "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]) : : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = q * 8 + (lane >> 3), ch = lane & 7;
      const int m = mbase + c * EP_ROWS + row;
      const uint4 dv = *reinterpret_cast<const uint4*>(strip + row * EP_PITCH_BF16 + ch * 16);
      const uint32_t dw[4] = {dv.x, dv.y, dv.z, dv.w};
      uint32_t og[4], ou[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float g0, g1, u0, u1;
        swiglu_bwd_elem(__uint_as_float(g[q][e] << 16), __uint_as_float(u[q][e] << 16), __uint_as_float(dw[e] << 16), g0, u0);
        swiglu_bwd_elem(__uint_as_float(g[q][e] & 0xffff0000u), __uint_as_float(u[q][e] & 0xffff0000u), __uint_as_float(dw[e] & 0xffff0000u), g1, u1);
        og[e] = pack_bf2(g0, g1); ou[e] = pack_bf2(u0, u1);
      }
      if (m < p.M && row < rows_here) {
        bf16_t* cp = C + (int64_t)m * p.ldc + nbase + ch * 8;
        UG_ST16(cp, make_uint4(og[0], og[1], og[2], og[3]));
        UG_ST16(cp + I, make_uint4(ou[0], ou[1], ou[2], ou[3]));
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

// DBUF = true : two LDS stages (64 KiB, 2 workgroups/CU), next tile's DMA overlaps this tile's MFMAs.
// DBUF = false: one LDS stage (32 KiB, up to 4 workgroups/CU), overlap comes from the other workgroups.
template <int EPI, bool AK, bool BKM, bool DBUF>
__global__ __launch_bounds__(256, DBUF ? 2 : 4) void gemm_kernel(GemmArgs p) {
  __shared__ __attribute__((aligned(16))) char lds[(DBUF ? 4 : 2) * TILE_BYTES];   // A0 B0 [A1 B1]
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

  Stager<AK, BM, 4> sa; Stager<BKM, BN, 4> sb;
  sa.init(p.A, p.lda, m0, p.M, wave, lane);
  sb.init(p.B, p.ldb, n0, p.N, wave, lane);

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nk_all = (p.K + BK - 1) / BK;
  const int per_split = (nk_all + gridDim.y - 1) / gridDim.y;       // split-K: this workgroup's k-tile range
  const int kt0 = blockIdx.y * per_split;
  const int nk = min(nk_all, kt0 + per_split);
  if (kt0 >= nk) return;
  const bool ragged = (p.K % BK) != 0;          // only the last k-tile can need the zero-page checks
  if (kt0 + 1 == nk_all && ragged) { sa.template issue<true>(kt0, p.K, lds, wave); sb.template issue<true>(kt0, p.K, lds + TILE_BYTES, wave); }
  else { sa.template issue<false>(kt0, p.K, lds, wave); sb.template issue<false>(kt0, p.K, lds + TILE_BYTES, wave); }
  __syncthreads();   // hipcc drains vmcnt(0) for in-flight LDS-DMA here

  for (int kt = kt0; kt < nk; ++kt) {
    const int cur = DBUF ? ((kt - kt0) & 1) : 0;
    if (DBUF && kt + 1 < nk) {
      char* nxt = lds + (cur ^ 1) * 2 * TILE_BYTES;
      if (ragged && kt + 2 == nk_all) { sa.template issue<true>(kt + 1, p.K, nxt, wave); sb.template issue<true>(kt + 1, p.K, nxt + TILE_BYTES, wave); }
      else { sa.template issue<false>(kt + 1, p.K, nxt, wave); sb.template issue<false>(kt + 1, p.K, nxt + TILE_BYTES, wave); }
    }
    const char* tA = lds + cur * 2 * TILE_BYTES;
    const char* tB = tA + TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fa[i] = load_frag<AK, BM>(tA, wm * 64 + i * 16, ks, lane);
        fb[i] = load_frag<BKM, BN>(tB, wn * 64 + i * 16, ks, lane);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)   // swapped operands: lane gets C[m=lane&15][n=(lane>>4)*4+r]
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
    if constexpr (!DBUF) {
      if (kt + 1 < nk) {
        if (ragged && kt + 2 == nk_all) { sa.template issue<true>(kt + 1, p.K, lds, wave); sb.template issue<true>(kt + 1, p.K, lds + TILE_BYTES, wave); }
        else { sa.template issue<false>(kt + 1, p.K, lds, wave); sb.template issue<false>(kt + 1, p.K, lds + TILE_BYTES, wave); }
        __syncthreads();
      }
    }
  }

  store_tile<EPI>(p, acc, m0 + wm * 64, n0 + wn * 64, lane, gridDim.y > 1);
}

// =============================================================================================
// Staggered two-group kernel: 256x256 tile, 8 waves = 2 groups x 4 waves, each wave 128x64
// (8x4 fragments), k-tiles of 32 in a 4-stage LDS ring (4 x 32 KiB), one workgroup per CU.
//
// Why: at 128x128 the LDS-DMA path (L2 -> LDS, ~17 TB/s measured) moves 15.6 B per kFLOP and is
// as long as the MFMA phase itself; a 256x256 tile halves that.  To keep the matrix pipe fed
// with one workgroup per CU, the two wave groups run the SAME loop one phase apart:
//     group 0:        L(t)  |  M(t)  |  L(t+1)  |  M(t+1) | ...          L = 12 ds_read_b128 + 4 LDS-DMA
//     group 1:   -    |  L(t)  |  M(t)  |  L(t+1)  | ...                 M = 32 MFMA (512 cycles)
// Every '|' is a workgroup barrier.  A CU's SIMDs each host one wave of either group, so while one
// runs its 32 MFMAs its partner does LDS reads and issues the DMA of tile t+3; the data it needs
// is never more than a counted `s_waitcnt vmcnt(8)` away (two newer DMA batches stay in flight).
// Ring safety: tile t is read in intervals 2t (group 0) and 2t+1 (group 1); its stage is refilled
// with tile t+4, issued no earlier than interval 2t+2 (inside L(t+1)).
constexpr int PBM = 256, PBN = 256, PBK = 32;
constexpr int P_TILE = PBM * PBK * 2;          // 16 KiB per operand per stage
constexpr int P_STAGE = 2 * P_TILE;            // 32 KiB
constexpr int P_NST = 4;
// Cache policy of the operand streams' LDS-DMA: the default (aux 0).  sc0 / nt / sc1 measured in round 4 (probe switches
// UG_CPOL_STAGER / UG_CPOL_P10A of tools/probes/probe_switches.patch), see docs/experiments.md.
constexpr int CPOL_STAGER = 0, CPOL_P10A = 0;
static_assert(CPOL_P10A == 0, "dma16 issues the default cache policy");

__device__ __forceinline__ int swz_rowk32(int row, int chunk) { return chunk ^ ((0 - (row >> 2)) & 3); }

template <bool KMAJOR>
struct Stager32 {                // 256-row x 32-k operand tile, 8 waves: 2 one-KiB DMA instructions per wave per k-tile
  const bf16_t* src[2];
  int64_t step[2];
  int kofs[2];
  // split_rows > 0 (row-major only): tile rows 0..127 are operand rows row0.., rows 128..255 are operand rows split_rows + row0..
  __device__ __forceinline__ void init(const bf16_t* X, int64_t ld, int row0, int rows_total, int wave, int lane, int split_rows = 0) {
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int inst = i * 8 + wave;
      if constexpr (!KMAJOR) {                 // [256 rows][4 chunks]: 16 rows per instruction
        const int row = inst * 16 + (lane >> 2);
        const int chunk = swz_rowk32(row, lane & 3);
        const int grow = (split_rows > 0 && row >= 128) ? split_rows + row0 + row - 128 : row0 + row;
        const int r = min(grow, rows_total - 1);
        src[i] = X + (int64_t)r * ld + chunk * 8;
        step[i] = PBK;
        kofs[i] = chunk * 8;
      } else {                                 // [32 k][32 chunks]: 2 k-rows per instruction
        const int k = inst * 2 + (lane >> 5);
        const int chunk = swz_krow(k, lane & 31);
        const int col = row0 + chunk * 8;
        const bool ok = (col + 8 <= ld);
        src[i] = ok ? X + (int64_t)k * ld + col : zero;
        step[i] = ok ? (int64_t)PBK * ld : 0;
        kofs[i] = k;
      }
    }
  }
  template <bool CHECK>
  __device__ __forceinline__ void issue(int kt, int K, char* lds_tile, int wave) const {
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bf16_t* s = src[i] + kt * step[i];
      if constexpr (CHECK) {
        const bool past = kt * PBK + kofs[i] >= (KMAJOR ? K : ((K + 7) & ~7));
        s = reinterpret_cast<const bf16_t*>(past ? reinterpret_cast<uintptr_t>(zero) : reinterpret_cast<uintptr_t>(s));
      }
      char* dst = lds_tile + (i * 8 + wave) * 1024;
      // Hand-issued (vmem_asm.h), not __builtin_amdgcn_global_load_lds: with the builtin hipcc's wait insertion puts s_waitcnt
      // vmcnt(0) in front of every ds_read_b64_tr_b16 of a k-major operand (it cannot tell the ring slot being read from the
      // slots being filled), so dgrad and the weight gradients drained the WHOLE four-slot ring in every k-iteration and the counted
      // vmcnt(8) below never held (round 6, ISA evidence: profiles/r06_vmcnt_findings.md).  The row-major loops were not affected.
      static_assert(CPOL_STAGER == 0, "dma16 issues the default cache policy");
      dma16(s, __builtin_amdgcn_readfirstlane(lds_addr_of(dst)));
    }
  }
};

template <bool KMAJOR>
__device__ __forceinline__ bf16x8_t load_frag32(const char* tile, int r0, int lane) {
  if constexpr (!KMAJOR) {
    const int row = r0 + (lane & 15);
    return *reinterpret_cast<const bf16x8_t*>(tile + row * 64 + swz_rowk32(row, lane >> 4) * 16);
  } else {
    const int i16 = lane & 15, g = lane >> 4;
    const int k = g * 8 + (i16 >> 2);
    const int chunk = (r0 >> 3) + ((i16 & 3) >> 1);
    const int sub = (i16 & 1) * 8;
    const char* p0 = tile + k * (PBM * 2) + swz_krow(k, chunk) * 16 + sub;
    const char* p1 = tile + (k + 4) * (PBM * 2) + swz_krow(k + 4, chunk) * 16 + sub;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p0);
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p1);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  }
}

#define P_BARRIER() asm volatile("s_barrier" ::: "memory")

// The MFMA block of one k-tile: acc[i][j] += A block i x B block j, issued serpentine (row block outer, columns alternately up and
// down: one operand unchanged between ANY two consecutive instructions).  Against row-block-outer and column-block-outer orders
// (probe switch UG_MFMA_ORDER of tools/probes/probe_switches.patch) -- measured at the power cap (tools/probes/ab.sh lib "ship mo1 mo4", two passes, TF/s): gate_up forward 1 275 / 1 268 -> 1 283 / 1 282,
// gate_up dgrad 1 394 / 1 392 -> 1 408 / 1 408, down dgrad 1 340 / 1 343 -> 1 354 / 1 366; in the step (three A/B pairs) GEMM launches
// -0.7 ... -0.9 ms.  Fewer operand-port toggles per instruction is fewer joules per flop, and joules are what the step is bound by.
#define UG_MFMA_BLOCK(NI)                                                                                                  \
  _Pragma("unroll") for (int i = 0; i < (NI); ++i) _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) {                       \
    const int j = (i & 1) ? 3 - jj : jj;                                                                                   \
    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);                                  \
    __builtin_amdgcn_sched_barrier(0); }

// the workgroup `bid` of one launch (or of one problem of a grouped launch)
template <int EPI, bool AK, bool BKM, bool ONEBAR>
__device__ __forceinline__ void p8_body(const GemmArgs& p, const int bid, char* lds) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wn = wave & 3;

  const int nwg = p.tiles_m * p.tiles_n;
  const bool tail = p.tail_split > 1 && bid >= p.full_tiles;
  const int tail_j = tail ? bid - p.full_tiles : 0;
  const int tile_lin = tail ? p.full_tiles + tail_j / p.tail_split : bid;
  const int pid = xcd_remap(tile_lin, nwg);
  const int per_group = P8_GROUP_M * p.tiles_n;
  const int gid = pid / per_group, first_m = gid * P8_GROUP_M;
  const int gsz = min(p.tiles_m - first_m, P8_GROUP_M);
  const int tm = first_m + (pid % per_group) % gsz;
  const int tn = (pid % per_group) / gsz;
  const int m0 = tm * PBM, n0 = tn * PBN;

  Stager32<AK> sa; Stager32<BKM> sb;
  sa.init(p.A, p.lda, m0, p.M, wave, lane);
  sb.init(p.B, p.ldb, n0, p.N, wave, lane);

  f32x4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nk_all = (p.K + PBK - 1) / PBK;
  const int per_split = tail ? (nk_all + p.tail_split - 1) / p.tail_split : nk_all;
  const int kt0 = tail ? (tail_j % p.tail_split) * per_split : 0;
  const int nk = min(nk_all, kt0 + per_split) - kt0;                // local k-tile count (ring slots are local indices)
  if (nk <= 0) return;
  const bool ragged = (p.K % PBK) != 0;
  auto stage_in = [&](int lt) {                 // 4 DMA instructions per wave
    const int kt = kt0 + lt;
    char* st = lds + (lt & (P_NST - 1)) * P_STAGE;
    if (ragged && kt + 1 == nk_all) { sa.template issue<true>(kt, p.K, st, wave); sb.template issue<true>(kt, p.K, st + P_TILE, wave); }
    else { sa.template issue<false>(kt, p.K, st, wave); sb.template issue<false>(kt, p.K, st + P_TILE, wave); }
  };
  // Steady state (t + 4 < nk): tile t+3 is not the launch's last k-tile -- it exists and is never the ragged one -- so the
  // iteration carries no scalar branch between the fragment reads and the barrier; the last four iterations take the checked
  // form.  (Three to four branches per k-tile cost 5-10 % of the loop: measured when a run-time prefetch distance was tried.)
  auto stage_whole = [&](int lt) {
    char* st = lds + (lt & (P_NST - 1)) * P_STAGE;
    sa.template issue<false>(kt0 + lt, p.K, st, wave);
    sb.template issue<false>(kt0 + lt, p.K, st + P_TILE, wave);
  };
  const int nk_steady = max(nk - 4, 0);
  bf16x8_t fa[8], fb[4];
  auto read_frags = [&](int t) {
    const char* tA = lds + (t & (P_NST - 1)) * P_STAGE;
    const char* tB = tA + P_TILE;
#pragma unroll
    for (int i = 0; i < 8; ++i) fa[i] = load_frag32<AK>(tA, grp * 128 + i * 16, lane);
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[j] = load_frag32<BKM>(tB, wn * 64 + j * 16, lane);
  };
  auto request = [&](int t, auto steady) {        // DMA of tile t+3
    if constexpr (decltype(steady)::value) stage_whole(t + 3);
    else if (t + 3 < nk) stage_in(t + 3);
  };
  auto landed = [&](int t, auto steady) {         // this wave's share of tile t+1 (the batches t+2, t+3 may stay in flight)
    if constexpr (decltype(steady)::value) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (t + 3 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (t + 2 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  auto mfmas = [&](int prio) {
    __builtin_amdgcn_sched_barrier(0);
    if (prio == 2) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1);
    UG_MFMA_BLOCK(8)
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };
  const std::true_type STEADY{};
  const std::false_type CHECKED{};
  // prologue: tiles 0..2 in flight, tile 0 landed
  stage_in(0);
  if (nk > 1) stage_in(1);
  if (nk > 2) stage_in(2);
  if (nk > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (nk > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if constexpr (ONEBAR) {
    // One workgroup barrier per k-tile.  Interval t (between barriers t and t+1): a group-0 wave runs L(t) then M(t), the
    // group-1 wave on the same SIMD runs M(t-1) (fragments kept in registers across the barrier) then L(t) -- the two halves
    // of the interval still pair one wave's 32 MFMAs with the other's fragment reads, but nobody waits at a mid-interval
    // barrier for the slower half.  Barrier t publishes tile t (every wave waited for its own DMA share at the end of
    // interval t-1) and frees the slot of tile t-1 (last read by group 1 in interval t-1) for the DMA of tile t+3.
    // Group 1's M phase opens the interval and must not be starved by group 0's (older waves win the MFMA arbiter at equal
    // priority: measured, group 1's 32 MFMAs stretched from 668 to 1152 clocks): it runs at the higher priority.
    if (grp == 0) {
      auto interval = [&](int t, auto steady) {
        P_BARRIER();
        read_frags(t);
        request(t, steady);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        mfmas(1);
        landed(t, steady);
      };
      int t = 0;
      for (; t < nk_steady; ++t) interval(t, STEADY);
      for (; t < nk; ++t) interval(t, CHECKED);
      P_BARRIER();                               // interval nk: group 1's last M phase
    } else {
      P_BARRIER();
      read_frags(0);
      request(0, CHECKED);
      landed(0, CHECKED);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      auto interval = [&](int t, auto steady) {
        P_BARRIER();
        mfmas(2);
        read_frags(t);
        request(t, steady);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        landed(t, steady);
      };
      int t = 1;
      for (; t < nk_steady; ++t) interval(t, STEADY);
      for (; t < nk; ++t) interval(t, CHECKED);
      P_BARRIER();
      mfmas(2);
    }
  } else {
    // Two barriers per k-tile: L(t) | M(t) with group 1 one phase behind group 0.
    P_BARRIER();
    if (grp == 1) P_BARRIER();                    // stagger: group 1 runs one phase behind group 0
    auto iteration = [&](int t, auto steady) {
      // ---------------- L phase: fragments of tile t into registers, DMA of tile t+3, retire tile t+1's DMA
      read_frags(t);
      request(t, steady);
      // this wave's share of tile t+1 must have landed before the barrier that opens the interval in which
      // group 0 reads it; newer batches (t+2, t+3) may stay in flight
      landed(t, steady);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      P_BARRIER();
      // ---------------- M phase
      mfmas(1);
      P_BARRIER();
    };
    int t = 0;
    for (; t < nk_steady; ++t) iteration(t, STEADY);
    for (; t < nk; ++t) iteration(t, CHECKED);
    if (grp == 0) P_BARRIER();                    // balance the barrier count of the staggered group
  }

  if (tail) {
    const int64_t slot = p.tail_private ? (int64_t)(tile_lin - p.full_tiles) * p.tail_split + tail_j % p.tail_split
                                        : (int64_t)(tile_lin - p.full_tiles);
    float* ws = p.tail_ws + slot * (PBM * PBN);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int ml = grp * 128 + i * 16 + (lane & 15);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float* dst = ws + ml * PBN + wn * 64 + j * 16 + (lane >> 4) * 4;
        if (p.tail_private) {
          *reinterpret_cast<float4*>(dst) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) atomicAdd(dst + r, acc[i][j][r]);
        }
      }
    }
    return;
  }
  // interior column panel with 16-byte-addressable rows: wide stores through the (now idle) ring; ragged panels keep the
  // guarded element-wise form
  const bool wide = (n0 + PBN <= p.N) && p.wide_epilogue &&
                    (EPI == EPI_BF16 ? (p.ldc % 8 == 0 && (!p.bias || (reinterpret_cast<uintptr_t>(p.bias) & 7) == 0))
                                     : (p.ldc % 4 == 0 && (EPI != EPI_RESID || p.ldr % 4 == 0)));
  if (wide) {                                    // (the balancing barrier above already put every wave past its last ring read)
    store_tile_lds<EPI>(p, acc, lds + wave * EP_STRIP, m0 + grp * 128, n0 + wn * 64, lane);
    return;
  }
  store_tile<EPI>(p, reinterpret_cast<f32x4_t (&)[4][4]>(acc[0]), m0 + grp * 128, n0 + wn * 64, lane, false);
  store_tile<EPI>(p, reinterpret_cast<f32x4_t (&)[4][4]>(acc[4]), m0 + grp * 128 + 64, n0 + wn * 64, lane, false);
}

template <int EPI, bool AK, bool BKM, bool ONEBAR = false>
__global__ __launch_bounds__(512, 2) void gemm_kernel_p8(GemmArgs p) {
  __shared__ __attribute__((aligned(16))) char lds[P_NST * P_STAGE];
  p8_body<EPI, AK, BKM, ONEBAR>(p, (int)blockIdx.x, lds);
}

// Grouped weight gradients: the four dW = dY^T X of a decoder layer (gate_up 420 tiles, down 210, qkv 48, o 36 at 12 336 tokens)
// share the contraction length, so their 256x256 tiles cost the same and ONE launch packs them into ceil(714 / 256) = 3 rounds
// (round 4: the contraction length is per problem -- the launch also carries a slice of the tied head's weight gradient, K = the
// label rows, whose short tiles run on the CUs the third round leaves idle)
// of the chip; launched one by one they take 2 + 1 rounds plus two k-sliced launches with their finishing passes.  Every
// problem's first block index is a multiple of 8 (XCD round-robin as in the single launch); padding blocks return at once.
constexpr int GROUP_MAX = 8;
struct GroupProblem {
  const bf16_t* A; const bf16_t* B; float* C;
  int64_t lda, ldb, ldc;
  int M, N, K, beta, tiles_m, tiles_n, start;
};
struct GroupArgs { GroupProblem pr[GROUP_MAX]; int n; };

__global__ __launch_bounds__(512, 2) void gemm_kernel_p8_wgrad_group(GroupArgs g) {
  __shared__ __attribute__((aligned(16))) char lds[P_NST * P_STAGE];
  const int b = (int)blockIdx.x;
  int q = 0;
#pragma unroll
  for (int i = 1; i < GROUP_MAX; ++i) if (i < g.n && b >= g.pr[i].start) q = i;
  const GroupProblem& pr = g.pr[q];
  const int local = b - pr.start;
  if (local >= pr.tiles_m * pr.tiles_n) return;
  GemmArgs p;
  p.A = pr.A; p.B = pr.B; p.C = pr.C; p.bias = nullptr; p.resid = nullptr; p.alpha_dev = nullptr;
  p.M = pr.M; p.N = pr.N; p.K = pr.K; p.lda = pr.lda; p.ldb = pr.ldb; p.ldc = pr.ldc; p.ldr = 0; p.beta = pr.beta;
  p.tiles_m = pr.tiles_m; p.tiles_n = pr.tiles_n; p.full_tiles = pr.tiles_m * pr.tiles_n; p.tail_split = 1;
  p.tail_ws = nullptr; p.tail_private = 1; p.one_barrier = 1; p.wide_epilogue = 1; p.swiglu_I = 0; p.act = nullptr; p.ld_act = 0;
  p8_body<EPI_F32, true, true, true>(p, local, lds);
}

// =============================================================================================
// 320 x 256 tiles for the token-count x 1536 outputs of the attention block (o forward, qkv / o dgrad: M = 12 336, K <= 2 048).
// Their 49 x 6 = 294 tiles of 256 x 256 are 1.15 rounds of the chip and the contraction is too short for k-slices; 39 x 6 = 234
// tiles of 320 x 256 are ONE round at 91 % occupancy.  Same two-barrier L | M schedule; a wave owns 160 x 64 (10 x 4 blocks,
// 40 MFMAs per k-tile, 160 accumulator registers).  Restricted to what those launches need: A row-major, K % 32 == 0,
// N % 256 == 0, bf16 / residual epilogues through the LDS strips, no k-slices.  The 20 A instructions of a k-tile go two
// per wave plus a third for the waves of group 0, so the counted waits are per group (5 / 4 DMA instructions per tile).
constexpr int P10_GROUP_M = 4;
constexpr int QBM = 320;
constexpr int Q_TILE_A = QBM * PBK * 2;        // 20 KiB
constexpr int Q_STAGE = Q_TILE_A + P_TILE;     // 36 KiB; four stages = 144 KiB

// Round 4: the row count of the tile is a template parameter -- group 0's waves own F0 row blocks of 16, group 1's F1 (F0 >= F1 >=
// F0 - 1): 16 (F0 + F1) rows, 320 at (10, 10).  A one-round launch takes as long as ONE tile, so the launcher picks the smallest tile
// height whose row tiles x column tiles still fit the 256 CUs: M = 12 336, N = 1 536 -> 304 rows (41 x 6 = 246 workgroups) instead of
// 320 (39 x 6 = 234): the same round, 5 % less work per workgroup.  A-tile DMA: 16 (F0 + F1) / 16 instructions of 16 rows per k-tile,
// dealt over the eight waves in order; a wave issues NA = LO or LO + 1 of them (counted waits per wave).
template <int EPI, bool BKM, int F0, int F1>
__global__ __launch_bounds__(512, 2) void gemm_kernel_p10(GemmArgs p) {
  constexpr int BMV = 16 * (F0 + F1), NINST = BMV / 16, LO = NINST / 8, REM = NINST % 8;
  constexpr int TILE_A = BMV * PBK * 2, STAGE = TILE_A + P_TILE;
  static_assert(F0 >= F1 && F0 <= 10 && F1 >= 1 && LO >= 1 && LO + (REM ? 1 : 0) <= 3, "gemm_kernel_p10: unsupported tile height");
  __shared__ __attribute__((aligned(16))) char lds[P_NST * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wn = wave & 3;
  const int nwg = p.tiles_m * p.tiles_n;
  const int pid = xcd_remap((int)blockIdx.x, nwg);
  const int per_group = P10_GROUP_M * p.tiles_n;
  const int gid = pid / per_group, first_m = gid * P10_GROUP_M;
  const int gsz = min(p.tiles_m - first_m, P10_GROUP_M);
  const int tm = first_m + (pid % per_group) % gsz;
  const int tn = (pid % per_group) / gsz;
  const int m0 = tm * BMV, n0 = tn * PBN;

  const bf16_t* asrc[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int inst = min(i * 8 + wave, NINST - 1);             // (slots past the tile are never issued)
    const int row = inst * 16 + (lane >> 2);
    const int r = min(m0 + row, p.M - 1);
    asrc[i] = p.A + (int64_t)r * p.lda + swz_rowk32(row, lane & 3) * 8;
  }
  Stager32<BKM> sb;
  if constexpr (EPI == EPI_SWIGLU) sb.init(p.B, p.ldb, tn * 128, p.N, wave, lane, p.swiglu_I);      // rows 0..127 gate units, 128..255 their up rows
  else sb.init(p.B, p.ldb, n0, p.N, wave, lane);

  f32x4_t acc[F0][4];
#pragma unroll
  for (int i = 0; i < F0; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int nk = p.K / PBK;

  auto run = [&](auto group0, auto na_tag) {
    constexpr bool G0 = decltype(group0)::value;
    constexpr int NA = decltype(na_tag)::value;                // A-tile DMA instructions of this wave per k-tile
    constexpr int FG = G0 ? F0 : F1;                           // row blocks of this wave
    constexpr int PER = NA + 2;                                // DMA instructions per batch (+ 2 of the B tile)
    auto stage_in = [&](int lt) {
      char* st = lds + (lt & (P_NST - 1)) * STAGE;
#pragma unroll
      for (int i = 0; i < NA; ++i)
        dma16(asrc[i] + (int64_t)lt * PBK, __builtin_amdgcn_readfirstlane(lds_addr_of(st + (i * 8 + wave) * 1024)));      // (see Stager32::issue)
      sb.template issue<false>(lt, p.K, st + TILE_A, wave);
    };
    auto landed = [&](int in_flight) {            // batches of this wave's DMA that may stay in flight
      if (in_flight >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
      else if (in_flight == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    stage_in(0);
    if (nk > 1) stage_in(1);
    if (nk > 2) stage_in(2);
    landed(min(nk, 3) - 1);
    P_BARRIER();
    if (!G0) P_BARRIER();                         // stagger: group 1 runs one phase behind group 0
    auto iteration = [&](int t, auto steady) {
      const char* tA = lds + (t & (P_NST - 1)) * STAGE;
      const char* tB = tA + TILE_A;
      bf16x8_t fa[FG], fb[4];
#pragma unroll
      for (int i = 0; i < FG; ++i) fa[i] = load_frag32<false>(tA, (G0 ? 0 : 16 * F0) + i * 16, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        fb[j] = load_frag32<BKM>(tB, EPI == EPI_ROPE ? (wn >> 1) * 128 + (wn & 1) * 32 + (j & 1) * 16 + (j >> 1) * 64
                                     : EPI == EPI_SWIGLU ? (j >> 1) * 128 + wn * 32 + (j & 1) * 16 : wn * 64 + j * 16, lane);
      if constexpr (decltype(steady)::value) {
        stage_in(t + 3);
        landed(2);
      } else {
        if (t + 3 < nk) stage_in(t + 3);
        landed(min(nk - 1, t + 3) - (t + 1));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      P_BARRIER();
      __builtin_amdgcn_s_setprio(1);
      UG_MFMA_BLOCK(FG)
      __builtin_amdgcn_s_setprio(0);
      P_BARRIER();
    };
    int t = 0;
    for (; t + 4 < nk; ++t) iteration(t, std::true_type{});
    for (; t < nk; ++t) iteration(t, std::false_type{});
    if (G0) P_BARRIER();
    if constexpr (EPI == EPI_SWIGLU_BWD)
      store_tile_swiglu_bwd<FG>(p, reinterpret_cast<f32x4_t (&)[FG][4]>(acc[0]), lds + wave * EP_STRIP, m0 + (G0 ? 0 : 16 * F0), n0 + wn * 64, lane);
    else if constexpr (EPI == EPI_SWIGLU)
      store_tile_swiglu_perm<FG>(p, reinterpret_cast<f32x4_t (&)[FG][4]>(acc[0]), lds + wave * EP_STRIP, m0 + (G0 ? 0 : 16 * F0), tn * 128 + wn * 32, lane);
    else if constexpr (EPI == EPI_ROPE)
      store_tile_lds<EPI, FG>(p, reinterpret_cast<f32x4_t (&)[FG][4]>(acc[0]), lds + wave * EP_STRIP, m0 + (G0 ? 0 : 16 * F0),
                              n0 + (wn >> 1) * 128 + (wn & 1) * 32, lane, (wn & 1) * 32, n0 < p.rope_cols);
    else
      store_tile_lds<EPI, FG>(p, reinterpret_cast<f32x4_t (&)[FG][4]>(acc[0]), lds + wave * EP_STRIP, m0 + (G0 ? 0 : 16 * F0), n0 + wn * 64, lane);
  };
  const bool more = REM != 0 && wave < REM;       // this wave issues LO + 1 A-tile instructions
  if (grp == 0) { if (more) run(std::true_type{}, std::integral_constant<int, LO + (REM ? 1 : 0)>{}); else run(std::true_type{}, std::integral_constant<int, LO>{}); }
  else { if (more) run(std::false_type{}, std::integral_constant<int, LO + (REM ? 1 : 0)>{}); else run(std::false_type{}, std::integral_constant<int, LO>{}); }
}


// Epilogue of the k-sliced tail tiles: scratch -> C with the launch's epilogue, scratch re-zeroed.
template <int EPI>
__global__ __launch_bounds__(256) void tail_finish_kernel(GemmArgs p, int ntail) {
  const int nwg = p.tiles_m * p.tiles_n;
  const int per_tile = PBM * PBN / 4;
  const int64_t total = (int64_t)ntail * per_tile;
  float alpha = 1.f;
  if constexpr (EPI == EPI_F32) { if (p.alpha_dev) alpha = *p.alpha_dev; }
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int tl = (int)(idx / per_tile), q = (int)(idx % per_tile);
    const int pid = xcd_remap(p.full_tiles + tl, nwg);
    const int per_group = P8_GROUP_M * p.tiles_n;
    const int gid = pid / per_group, first_m = gid * P8_GROUP_M;
    const int gsz = min(p.tiles_m - first_m, P8_GROUP_M);
    const int tm = first_m + (pid % per_group) % gsz;
    const int tn = (pid % per_group) / gsz;
    const int ml = q / (PBN / 4), nl = (q % (PBN / 4)) * 4;
    float4 v4;
    if (p.tail_private) {      // sum of the k-slices' private partials (a slice past K wrote nothing: skipped)
      const int nk_all = (p.K + PBK - 1) / PBK, per_split = (nk_all + p.tail_split - 1) / p.tail_split;
      v4 = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int sp = 0; sp < p.tail_split && sp * per_split < nk_all; ++sp) {
        const float4 t = *reinterpret_cast<const float4*>(p.tail_ws + ((int64_t)tl * p.tail_split + sp) * (PBM * PBN) + ml * PBN + nl);
        v4.x += t.x; v4.y += t.y; v4.z += t.z; v4.w += t.w;
      }
    } else {
      float4* src = reinterpret_cast<float4*>(p.tail_ws + (int64_t)tl * (PBM * PBN) + ml * PBN + nl);
      v4 = *src;
      *src = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int m = tm * PBM + ml, n = tn * PBN + nl;
    if (m >= p.M || n >= p.N) continue;
    float v[4] = {v4.x, v4.y, v4.z, v4.w};
    for (int r = 0; r < 4 && n + r < p.N; ++r) {
      if constexpr (EPI == EPI_BF16) {
        float o = v[r];
        if (p.bias) o += bf2f(p.bias[n + r]);
        reinterpret_cast<bf16_t*>(p.C)[(int64_t)m * p.ldc + n + r] = f2bf(o);
      } else if constexpr (EPI == EPI_F32) {
        float* c = reinterpret_cast<float*>(p.C) + (int64_t)m * p.ldc + n + r;
        *c = (p.beta ? *c : 0.f) + v[r] * alpha;
      } else {
        reinterpret_cast<float*>(p.C)[(int64_t)m * p.ldc + n + r] = p.resid[(int64_t)m * p.ldr + n + r] + bf2f(f2bf(v[r]));
      }
    }
  }
}

// Scratch of the k-sliced tiles: WS_PRIVATE_SLOTS slots of 256 KiB for private partials (192 MiB: room for 3 rounds of
// k-slices).  It belongs to a ug_handle (ug_create / ug_destroy, runtime.hip): nothing on the op path allocates, clears or
// synchronises, and two streams that each use their own handle never share partials.  A launch without a handle simply
// does not use the k-sliced forms.
constexpr int WS_PRIVATE_SLOTS = UG_HANDLE_WS_SLOTS;

template <int EPI, bool AK, bool BKM>
int launch(GemmArgs a, const ug_handle* h, int policy, hipStream_t st) {
  float* const ws = (h && h->tail_ws) ? h->tail_ws : nullptr;          // k-sliced forms need a handle's scratch
  const int g_tile_policy = policy;                                    // per call: -1 auto, see ug_gemm_bf16 in the header
  const int tiles = a.tiles_m * a.tiles_n;
  int splits = 1;
  if (EPI == EPI_F32 && a.beta == 1 && tiles < 384) {            // wgrad of a small weight: fill the chip along K
    const int nk = (a.K + BK - 1) / BK;
    splits = min(32, max(1, 768 / tiles));                         // skinny decode GEMMs (M = 16) get up to 32 slices
    splits = max(1, min(splits, nk / 2));
  }
  // Staggered 256x256 kernel when its one-workgroup-per-CU grid quantises well (measured, tools/gemm_bench.py:
  // +8..17 % on the wide shapes; the 294-tile N=1536 shapes lose a half-empty second round and stay on 128x128).
  const int tiles_p8 = ((a.M + PBM - 1) / PBM) * ((a.N + PBN - 1) / PBN);
  const int rounds = (tiles_p8 + 255) / 256;
  const bool p8_fits = (tiles_p8 <= 256) ? (tiles_p8 >= 200) : (4 * tiles_p8 >= 3 * rounds * 256);
  // token-count x 1536 outputs (o / down forward, qkv / o / gate_up dgrad): one round of 320 x 256 tiles instead of 1.15 rounds
  // of 256 x 256 -- also ahead of the k-sliced tail where the contraction is long (down forward 954 -> 1065, gate_up dgrad
  // 1188 -> 1354 TF/s)
  if constexpr (!AK && EPI != EPI_F32) {
    const int tiles_q = ((a.M + QBM - 1) / QBM) * (a.N / PBN);
    const bool aligned = a.N % PBN == 0 && a.K % PBK == 0 &&
                         (EPI == EPI_BF16 ? (a.ldc % 8 == 0 && (!a.bias || (reinterpret_cast<uintptr_t>(a.bias) & 7) == 0))
                                          : (a.ldc % 4 == 0 && a.ldr % 4 == 0));
    // many rounds: a 320-row tile costs 1.25 x a 256-row one and runs ~7 % more efficiently (fewer LDS and DMA bytes per MFMA);
    // taken when whole rounds come out at least 3 % cheaper (gate_up forward: 14 rounds -> 11 x 1.16 = 12.8: 1240 -> 1322 TF/s)
    const bool fewer_rounds = tiles_p8 >= 1024 && 1.1625f * (float)((tiles_q + 255) / 256) < 0.97f * (float)((tiles_p8 + 255) / 256);
    // Round 4: the same comparison over every tile height -- rounds(h) x t(h), t(h) = 0.35 + 0.00254 h in units of a 256-row tile
    // (from the two measured points t(256) = 1, t(320) = 1.1625).  down dgrad (12 336 x 8 960): 288 rows = 6 rounds x 1.08 against 7
    // rounds of 256 (286 -> 255 us); 9 288 rows: 272; gate_up forward keeps 320 at 12 336 rows and takes 304 at 9 288
    // (tools/gemm_shape_sweep.py agrees with the model's ranking on every shape it changes).
    int hb_multi = 0;
    // One-round launches (round 4): a round takes as long as one tile, so the SMALLEST tile height whose row tiles x column tiles
    // still fit the 256 CUs wins -- 12 336 x 1 536 outputs: 304 rows (246 workgroups) instead of 320 (234); 9 288 rows: 224.
    int hb = 0;
    static const bool heights_on = !(getenv("UNIGEN_GEMM_HEIGHTS") && atoi(getenv("UNIGEN_GEMM_HEIGHTS")) == 0);   // A/B switch
    if (aligned && g_tile_policy < 0 && a.N / PBN >= 1 && a.N / PBN <= 64) {
      static const int heights[] = {128, 144, 160, 176, 192, 208, 224, 240, 256, 272, 288, 304, 320};
      for (int h_ : heights) {
        if (!heights_on && h_ != QBM) continue;                  // (round 3's choice: 320 rows or nothing)
        const int64_t wgs = (int64_t)((a.M + h_ - 1) / h_) * (a.N / PBN);
        // (measured down to 78 workgroups: tools/gemm_shape_sweep.py at M = 1 542.)  Few workgroups AND a very long contraction
        // (an lm-head dgrad whose vocabulary is a multiple of 32: >= 2048 k-tiles on < 200 CUs) stay with the k-sliced form
        // below, which cuts every tile along K over the whole chip (measured 632 -> 1 094 TF/s at 174 tiles, K = 159 867)
        if (wgs <= 256) { if (wgs >= 64 && !(a.K / PBK >= 2048 && wgs < 200)) hb = h_; break; }
      }
      if (hb == 256 && p8_fits) hb = 0;                          // (the 256 x 256 kernel's own one-round case)
    }
    if (aligned && g_tile_policy < 0 && heights_on && hb == 0 && tiles_p8 >= 768) {
      static const int mheights[] = {176, 192, 208, 224, 240, 272, 288, 304, 320};
      float best = 0.97f * (float)((tiles_p8 + 255) / 256);
      for (int h_ : mheights) {
        const int64_t wgs = (int64_t)((a.M + h_ - 1) / h_) * (a.N / PBN);
        const float cost = (float)((wgs + 255) / 256) * (0.35f + 0.00254f * (float)h_);
        if (cost < best) { best = cost; hb_multi = h_; }
      }
    }
    if (aligned && g_tile_policy >= 40 && g_tile_policy <= 52 && g_tile_policy != 48) hb = 16 * (g_tile_policy - 32);   // forced height (tests, A/B)
    if (hb == 0 && hb_multi != 0) hb = hb_multi;
    if (hb == 0 && aligned && (g_tile_policy == 10 || (g_tile_policy < 0 && !heights_on && fewer_rounds))) hb = QBM;
    if (hb != 0) {
      a.tiles_m = (a.M + hb - 1) / hb; a.tiles_n = a.N / PBN;
      const dim3 grid(a.tiles_m * a.tiles_n), block(512);
      switch (hb) {
        case 320: hipLaunchKernelGGL((gemm_kernel_p10<EPI, BKM, 10, 10>), grid, block, 0, st, a); break;
        case 304: hipLaunchKernelGGL((gemm_kernel_p10<EPI, BKM, 10, 9>), grid, block, 0, st, a); break;
        case 288: hipLaunchKernelGGL((gemm_kernel_p10<EPI, BKM, 9, 9>), grid, block, 0, st, a); break;
        case 272: hipLaunchKernelGGL((gemm_kernel_p10<EPI, BKM, 9, 8>), grid, block, 0, st, a); break;
        case 256: hipLaunchKernelGGL((gemm_kernel_p10<EPI, BKM, 8, 8>), grid, block, 0, st, a); break;
        case 240: hipLaunchKernelGGL((gemm_kernel_p10<EPI, BKM, 8, 7>), grid, block, 0, st, a); break;
        case 224: hipLaunchKernelGGL((gemm_kernel_p10<EPI, BKM, 7, 7>), grid, block, 0, st, a); break;
        case 208: hipLaunchKernelGGL((gemm_kernel_p10<EPI, BKM, 7, 6>), grid, block, 0, st, a); break;
        case 192: hipLaunchKernelGGL((gemm_kernel_p10<EPI, BKM, 6, 6>), grid, block, 0, st, a); break;
        case 176: hipLaunchKernelGGL((gemm_kernel_p10<EPI, BKM, 6, 5>), grid, block, 0, st, a); break;
        case 160: hipLaunchKernelGGL((gemm_kernel_p10<EPI, BKM, 5, 5>), grid, block, 0, st, a); break;
        case 144: hipLaunchKernelGGL((gemm_kernel_p10<EPI, BKM, 5, 4>), grid, block, 0, st, a); break;
        default: hipLaunchKernelGGL((gemm_kernel_p10<EPI, BKM, 4, 4>), grid, block, 0, st, a); break;
      }
      UG_CHECK_LAUNCH("ug_gemm_bf16(p10)");
      return UG_OK;
    }
  }
  // A partial last round of 256x256 tiles is cut along K instead: r = tiles mod 256 leftover tiles x s slices fill
  // the chip once more for 1/s of a tile time (fp32 atomics into a scratch, then tail_finish applies the epilogue).
  int tail_r = 0, tail_s = 1;
  if (!p8_fits && tiles_p8 > 256 && (tiles_p8 % 256) <= 128 && g_tile_policy != 5) {
    const int r = tiles_p8 % 256, nk32 = (a.K + PBK - 1) / PBK;
    int sp = 256 / r;
    // measured with private partials (tools/gemm_bench.py): slices of >= 40 k-tiles and >= 3 slices win (down fwd
    // 856 -> 1073, gate_up dgrad 748 -> ~1000 TF/s); the K = 1536 shapes (<= 24 k-tiles per slice) lose to 128x128
    while (sp > 1 && nk32 / sp < 40) --sp;
    if ((sp >= 3 || g_tile_policy == 6) && sp >= 2 && r * sp <= WS_PRIVATE_SLOTS && ws) { tail_r = r; tail_s = sp; }
  }
  // Small accumulating fp32 outputs (weight gradients of the attention projections: 48 / 36 tiles): cut EVERY tile
  // along K so one round fills the chip, each slice storing a private partial that a finishing pass sums into C.
  // (The same cut with fp32 atomics was 118-316 TF/s, the 128x128 kernel's 4 atomic slices 353: atomics cost more
  // than the GEMM itself at these sizes.)
  {
    const int nk32 = (a.K + PBK - 1) / PBK;
    int sp = 0;
    if (EPI == EPI_F32 && tiles_p8 >= 24 && tiles_p8 <= 128 && a.M >= 512 && a.N >= 512) {
      sp = 256 / tiles_p8;                               // one round
      while (sp > 1 && nk32 / sp < 32) --sp;
    } else if (tiles_p8 >= 24 && tiles_p8 < 200 && nk32 >= 2048) {
      // few output tiles, very long contraction (lm-head dgrad: 96 tiles, K = 159 867): up to three rounds of slices,
      // the count that fills whole rounds best (632 -> 1094 TF/s); up to 199 tiles since round 4 (the pt1 mixed batch's
      // 7 184 head rows = 174 tiles ran 2.7 rounds of 128 x 128 tiles at 817 TF/s)
      float best = (tiles_p8 > 128) ? (float)tiles_p8 / 256.f : 0.f;      // (must beat the unsliced single round)
      for (int c = 2; c <= 8 && tiles_p8 * c <= WS_PRIVATE_SLOTS; ++c) {
        const int items = tiles_p8 * c, r = (items + 255) / 256;
        const float eff = (float)items / (float)(r * 256);
        if (eff > best + 0.01f) { best = eff; sp = c; }
      }
    }
    if (sp >= 2 && tiles_p8 * sp <= WS_PRIVATE_SLOTS && (g_tile_policy < 0 || g_tile_policy == 8) && ws) {
      a.tiles_m = (a.M + PBM - 1) / PBM; a.tiles_n = (a.N + PBN - 1) / PBN;
      a.full_tiles = 0; a.tail_split = sp; a.tail_private = 1;
      a.tail_ws = ws;
      if (a.one_barrier) hipLaunchKernelGGL((gemm_kernel_p8<EPI, AK, BKM, true>), dim3(tiles_p8 * sp), dim3(512), 0, st, a);
      else hipLaunchKernelGGL((gemm_kernel_p8<EPI, AK, BKM>), dim3(tiles_p8 * sp), dim3(512), 0, st, a);
      UG_CHECK_LAUNCH("ug_gemm_bf16(p8 k-sliced)");
      hipLaunchKernelGGL((tail_finish_kernel<EPI>), dim3(tiles_p8 * 16), dim3(256), 0, st, a, tiles_p8);
      UG_CHECK_LAUNCH("ug_gemm_bf16(partial sum)");
      return UG_OK;
    }
  }
  if (g_tile_policy == 3 || g_tile_policy == 6 || (g_tile_policy < 0 && (p8_fits || tail_s > 1))) {
    a.tiles_m = (a.M + PBM - 1) / PBM; a.tiles_n = (a.N + PBN - 1) / PBN;
    a.full_tiles = tiles_p8 - tail_r; a.tail_split = tail_s;
    a.tail_private = 1;                          // private partials beat atomics here too (gate_up dgrad 886 -> see DESIGN)
    a.tail_ws = ws;
    if (a.one_barrier) hipLaunchKernelGGL((gemm_kernel_p8<EPI, AK, BKM, true>), dim3(a.full_tiles + tail_r * tail_s), dim3(512), 0, st, a);
    else hipLaunchKernelGGL((gemm_kernel_p8<EPI, AK, BKM>), dim3(a.full_tiles + tail_r * tail_s), dim3(512), 0, st, a);
    UG_CHECK_LAUNCH("ug_gemm_bf16(p8)");
    if (tail_s > 1) {
      hipLaunchKernelGGL((tail_finish_kernel<EPI>), dim3(tail_r * 16), dim3(256), 0, st, a, tail_r);
      UG_CHECK_LAUNCH("ug_gemm_bf16(tail finish)");
    }
    return UG_OK;
  }
  bool dbuf = (!AK && !BKM && a.K >= 4096);
  if (g_tile_policy == 0) dbuf = true;
  if (g_tile_policy == 2) dbuf = false;
  dim3 grid(tiles, splits);
  if (dbuf) hipLaunchKernelGGL((gemm_kernel<EPI, AK, BKM, true>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((gemm_kernel<EPI, AK, BKM, false>), grid, dim3(256), 0, st, a);
  UG_CHECK_LAUNCH("ug_gemm_bf16");
  return UG_OK;
}

}  // namespace

extern "C" int ug_gemm_bf16(const ug_handle* h, const void* A, int64_t lda, int a_kmajor, const void* B, int64_t ldb, int b_kmajor,
                            void* C, int64_t ldc, int64_t M, int64_t N, int64_t K, int epilogue, const void* bias,
                            const float* resid, int64_t ldr, int beta, const float* alpha_dev, int policy, hipStream_t stream) {
  UG_REQUIRE(M > 0 && N > 0 && K > 0, "ug_gemm_bf16: empty problem M=%ld N=%ld K=%ld", (long)M, (long)N, (long)K);
  UG_REQUIRE(lda % 8 == 0 && ldb % 8 == 0, "ug_gemm_bf16: lda/ldb must be multiples of 8 elements (16-byte rows)");
  UG_REQUIRE(ug_aligned16(A) && ug_aligned16(B) && ug_aligned16(C), "ug_gemm_bf16: A/B/C must be 16-byte aligned");
  UG_REQUIRE(ldc % 4 == 0, "ug_gemm_bf16: ldc must be a multiple of 4 elements");
  UG_REQUIRE(M < (1 << 30) && N < (1 << 30) && K < (1 << 30), "ug_gemm_bf16: dims too large");
  // K needs no tile alignment: lanes past the contraction extent fetch from a zero page.  A row-major
  // operand is fetched in 8-element chunks, so when K % 8 != 0 the chunk straddling K is read whole: the
  // other operand must then be k-major (its rows >= K come from the zero page) and the straddling tail
  // must hold finite values (the callers keep it zero).
  UG_REQUIRE(a_kmajor || b_kmajor || K % 8 == 0,
             "ug_gemm_bf16: K=%ld must be a multiple of 8 when both operands are row-major", (long)K);
  UG_REQUIRE(a_kmajor || lda >= ((K + 7) & ~7LL), "ug_gemm_bf16: row-major A needs lda >= round_up(K,8)");
  UG_REQUIRE(b_kmajor || ldb >= ((K + 7) & ~7LL), "ug_gemm_bf16: row-major B needs ldb >= round_up(K,8)");
  UG_REQUIRE(!a_kmajor || lda >= M, "ug_gemm_bf16: k-major A needs lda >= M");
  UG_REQUIRE(!b_kmajor || ldb >= N, "ug_gemm_bf16: k-major B needs ldb >= N");
  GemmArgs a;
  a.A = (const bf16_t*)A; a.B = (const bf16_t*)B; a.C = C;
  a.bias = (const bf16_t*)bias; a.resid = resid; a.alpha_dev = alpha_dev;
  a.M = (int)M; a.N = (int)N; a.K = (int)K;
  a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.ldr = ldr; a.beta = beta;
  a.swiglu_I = 0; a.act = nullptr; a.ld_act = 0;
  a.wide_epilogue = (policy >= 0 && (policy & UG_GEMM_NARROW_EPILOGUE)) ? 0 : 1;
  // main loop of the 256x256 kernel: one barrier per k-tile for the weight gradients (both operands through the transposing
  // fragment reads make L half again as long as M: +10..13 %), two for forward and dgrad (short L: one barrier is -2..-3 %)
  a.one_barrier = (a_kmajor && b_kmajor) ? 1 : 0;
  if (policy >= 0 && (policy & UG_GEMM_ONE_BARRIER)) a.one_barrier = 1;
  if (policy >= 0 && (policy & UG_GEMM_TWO_BARRIERS)) a.one_barrier = 0;
  if (policy >= 0) policy &= ~(UG_GEMM_NARROW_EPILOGUE | UG_GEMM_ONE_BARRIER | UG_GEMM_TWO_BARRIERS);
  if (policy == UG_GEMM_POLICY_AUTO_BITS) policy = -1;
  a.tiles_m = (int)((M + BM - 1) / BM); a.tiles_n = (int)((N + BN - 1) / BN);
  const int mode = (a_kmajor ? 2 : 0) | (b_kmajor ? 1 : 0);
  if (epilogue == EPI_RESID) UG_REQUIRE(resid != nullptr && ldr % 4 == 0, "ug_gemm_bf16: EPI_RESID needs a 16B-aligned residual");
  switch (mode * 4 + epilogue) {
    case 0 * 4 + EPI_BF16: return launch<EPI_BF16, false, false>(a, h, policy, stream);
    case 0 * 4 + EPI_F32: return launch<EPI_F32, false, false>(a, h, policy, stream);
    case 0 * 4 + EPI_RESID: return launch<EPI_RESID, false, false>(a, h, policy, stream);
    case 1 * 4 + EPI_BF16: return launch<EPI_BF16, false, true>(a, h, policy, stream);
    case 1 * 4 + EPI_F32: return launch<EPI_F32, false, true>(a, h, policy, stream);
    case 3 * 4 + EPI_F32: return launch<EPI_F32, true, true>(a, h, policy, stream);
    case 3 * 4 + EPI_BF16: return launch<EPI_BF16, true, true>(a, h, policy, stream);
    default:
      ug_set_error("ug_gemm_bf16: layout/epilogue combination (a_kmajor=%d b_kmajor=%d epilogue=%d) is not instantiated",
                   a_kmajor, b_kmajor, epilogue);
      return UG_ERR_ARG;
  }
}

extern "C" int ug_gemm_bf16_wgrad_group(int n, const void* const* dy, const int64_t* ld_dy, const void* const* x, const int64_t* ld_x,
                                        float* const* dw, const int64_t* ld_dw, const int64_t* rows, const int64_t* cols,
                                        const int* beta, const int64_t* K, hipStream_t stream) {
  UG_REQUIRE(n >= 1 && n <= GROUP_MAX && K, "ug_gemm_bf16_wgrad_group: 1..%d problems and their contraction lengths required", GROUP_MAX);
  GroupArgs g;
  g.n = n;
  int start = 0;
  for (int i = 0; i < n; ++i) {
    UG_REQUIRE(rows[i] > 0 && cols[i] > 0 && rows[i] < (1 << 30) && cols[i] < (1 << 30) && K[i] > 0 && K[i] < (1 << 30),
               "ug_gemm_bf16_wgrad_group: empty problem %d", i);
    UG_REQUIRE(ld_dy[i] % 8 == 0 && ld_x[i] % 8 == 0 && ld_dy[i] >= rows[i] && ld_x[i] >= cols[i] && ld_dw[i] % 4 == 0 && ld_dw[i] >= cols[i],
               "ug_gemm_bf16_wgrad_group: problem %d: token-major operands need leading dimensions that are multiples of 8 and cover "
               "their rows; ld_dw a multiple of 4", i);
    UG_REQUIRE(ug_aligned16(dy[i]) && ug_aligned16(x[i]) && ug_aligned16(dw[i]), "ug_gemm_bf16_wgrad_group: problem %d: 16-byte alignment", i);
    GroupProblem& pr = g.pr[i];
    pr.A = (const bf16_t*)dy[i]; pr.B = (const bf16_t*)x[i]; pr.C = dw[i];
    pr.lda = ld_dy[i]; pr.ldb = ld_x[i]; pr.ldc = ld_dw[i];
    pr.M = (int)rows[i]; pr.N = (int)cols[i]; pr.K = (int)K[i]; pr.beta = beta[i];
    pr.tiles_m = (pr.M + PBM - 1) / PBM; pr.tiles_n = (pr.N + PBN - 1) / PBN;
    pr.start = start;
    start += (pr.tiles_m * pr.tiles_n + 7) & ~7;
  }
  for (int i = n; i < GROUP_MAX; ++i) { g.pr[i] = g.pr[0]; g.pr[i].start = 1 << 30; }
  hipLaunchKernelGGL(gemm_kernel_p8_wgrad_group, dim3(start), dim3(512), 0, stream, g);
  UG_CHECK_LAUNCH("ug_gemm_bf16_wgrad_group");
  return UG_OK;
}


// Tile height of the fused-epilogue launches (ug_gemm_bf16_swiglu / _qkv_rope / _swiglu_bwd): rounds(h) x t(h) over the heights the
// entry point instantiates -- or the height ug_gemm_set_fused_tile_height() forces (tests and A/B runs reach every instantiation
// that way; 0 = automatic).  A forced height the entry point does not instantiate is refused.
static std::atomic<int> g_fused_height{0};
extern "C" int ug_gemm_set_fused_tile_height(int rows) {
  UG_REQUIRE(rows == 0 || (rows >= 128 && rows <= 320 && rows % 16 == 0), "ug_gemm_set_fused_tile_height: 0 (automatic) or 128 ... 320 in steps of 16 (%d)", rows);
  g_fused_height.store(rows);
  return UG_OK;
}
template <size_t NH>
static int pick_fused_height(const int (&heights)[NH], int64_t M, int64_t col_tiles, const char* who, int* out) {
  const int forced = g_fused_height.load();
  if (forced) {
    for (int h_ : heights)
      if (h_ == forced) { *out = forced; return UG_OK; }
    ug_set_error("%s: tile height %d forced by ug_gemm_set_fused_tile_height is not instantiated for this epilogue", who, forced);
    return UG_ERR_ARG;
  }
  int hb = heights[NH - 1]; float best = 1e30f;
  for (int h_ : heights) {
    const int64_t wgs = ((M + h_ - 1) / h_) * col_tiles;
    const float cost = (float)((wgs + 255) / 256) * (0.35f + 0.00254f * (float)h_);
    if (cost < best) { best = cost; hb = h_; }
  }
  *out = hb;
  return UG_OK;
}

extern "C" int ug_swiglu_fwd(const void* gate_up, void* act, int64_t tokens, int64_t I, hipStream_t st);

extern "C" int ug_gemm_bf16_swiglu(const ug_handle* h, const void* x, int64_t ldx, const void* w_gate_up, int64_t ldw, void* gu,
                                   int64_t ld_gu, void* act, int64_t ld_act, int64_t M, int64_t I, int64_t K, hipStream_t stream) {
  UG_REQUIRE(M > 0 && I > 0 && K > 0 && K % 8 == 0 && I % 8 == 0, "ug_gemm_bf16_swiglu: bad problem M=%ld I=%ld K=%ld", (long)M, (long)I, (long)K);
  UG_REQUIRE(ldx % 8 == 0 && ldw % 8 == 0 && ld_gu % 8 == 0 && ld_act % 8 == 0 && ldx >= K && ldw >= K && ld_gu >= 2 * I && ld_act >= I,
             "ug_gemm_bf16_swiglu: row strides must be multiples of 8 elements and cover the rows");
  UG_REQUIRE(ug_aligned16(x) && ug_aligned16(w_gate_up) && ug_aligned16(gu) && ug_aligned16(act), "ug_gemm_bf16_swiglu: alignment");
  UG_REQUIRE(M < (1 << 30) && I < (1 << 29) && K < (1 << 30), "ug_gemm_bf16_swiglu: dims too large");
  const int64_t tiles = ((M + PBM - 1) / PBM) * (2 * I / PBN);
  if (I % 128 != 0 || tiles < 64 || K % PBK != 0 || 2 * I / PBN > (1 << 20)) {
    // shapes the 256x256 kernel is not chosen for: the projection and the activation as two launches (identical values)
    if (int rc = ug_gemm_bf16(h, x, ldx, 0, w_gate_up, ldw, 0, gu, ld_gu, M, 2 * I, K, EPI_BF16, nullptr, nullptr, 0, 0, nullptr, -1, stream))
      return rc;
    UG_REQUIRE(ld_gu == 2 * I && ld_act == I, "ug_gemm_bf16_swiglu: the two-launch form needs packed gu / act rows");
    return ug_swiglu_fwd(gu, act, M, I, stream);
  }
  GemmArgs a{};
  a.A = (const bf16_t*)x; a.B = (const bf16_t*)w_gate_up; a.C = gu;
  a.M = (int)M; a.N = (int)(2 * I); a.K = (int)K;
  a.lda = ldx; a.ldb = ldw; a.ldc = ld_gu;
  a.tail_split = 1; a.tail_private = 1; a.wide_epilogue = 1;
  a.swiglu_I = (int)I; a.act = (bf16_t*)act; a.ld_act = ld_act;
  // round 4: the 128 ... 320-row kernel with gate and up of a hidden unit in one lane (EPI_SWIGLU); tile height by rounds(h) x t(h)
  static const int heights[] = {128, 160, 192, 208, 224, 256, 288, 320};
  int hb = 320;
  if (int rc = pick_fused_height(heights, M, 2 * I / PBN, "ug_gemm_bf16_swiglu", &hb)) return rc;
  a.tiles_m = (int)((M + hb - 1) / hb); a.tiles_n = (int)(2 * I / PBN);
  const dim3 grid(a.tiles_m * a.tiles_n), block(512);
  switch (hb) {
    case 320: hipLaunchKernelGGL((gemm_kernel_p10<EPI_SWIGLU, false, 10, 10>), grid, block, 0, stream, a); break;
    case 288: hipLaunchKernelGGL((gemm_kernel_p10<EPI_SWIGLU, false, 9, 9>), grid, block, 0, stream, a); break;
    case 256: hipLaunchKernelGGL((gemm_kernel_p10<EPI_SWIGLU, false, 8, 8>), grid, block, 0, stream, a); break;
    case 224: hipLaunchKernelGGL((gemm_kernel_p10<EPI_SWIGLU, false, 7, 7>), grid, block, 0, stream, a); break;
    case 208: hipLaunchKernelGGL((gemm_kernel_p10<EPI_SWIGLU, false, 7, 6>), grid, block, 0, stream, a); break;
    case 192: hipLaunchKernelGGL((gemm_kernel_p10<EPI_SWIGLU, false, 6, 6>), grid, block, 0, stream, a); break;
    case 160: hipLaunchKernelGGL((gemm_kernel_p10<EPI_SWIGLU, false, 5, 5>), grid, block, 0, stream, a); break;
    default: hipLaunchKernelGGL((gemm_kernel_p10<EPI_SWIGLU, false, 4, 4>), grid, block, 0, stream, a); break;
  }
  UG_CHECK_LAUNCH("ug_gemm_bf16_swiglu(p10)");
  return UG_OK;
}

// The fused q/k/v projection: qkv = bf16(x W^T + b) with rotate-half RoPE applied to the first rope_cols columns (the q and k heads)
// in the epilogue (EPI_ROPE above) -- the values of ug_gemm_bf16 followed by ug_rope, bit for bit, without the in-place pass over
// the q / k columns (88 MB read + write per layer at the benchmark shape, 16.8 us).  transformers modeling_qwen2.py:131-135, 200-215.
extern "C" int ug_rope(void* qkv, const float* cos_tab, const float* sin_tab, int64_t tokens, int64_t L, int64_t ldq,
                       int nheads, int head_dim, int backward, hipStream_t st);

extern "C" int ug_gemm_bf16_qkv_rope(const ug_handle* h, const void* x, int64_t ldx, const void* w, int64_t ldw, const void* bias, void* qkv,
                                     int64_t ldq, int64_t M, int64_t N, int64_t K, const float* cos_tab, const float* sin_tab, int64_t L,
                                     int64_t rope_cols, int head_dim, hipStream_t stream) {
  UG_REQUIRE(M > 0 && N > 0 && K > 0 && L > 0 && M % L == 0 && head_dim > 0 && rope_cols >= 0 && rope_cols <= N && rope_cols % head_dim == 0,
             "ug_gemm_bf16_qkv_rope: bad problem M=%ld N=%ld K=%ld L=%ld rope_cols=%ld head_dim=%d", (long)M, (long)N, (long)K, (long)L,
             (long)rope_cols, head_dim);
  UG_REQUIRE(cos_tab && sin_tab && ug_aligned16(cos_tab) && ug_aligned16(sin_tab), "ug_gemm_bf16_qkv_rope: cos / sin tables [L, head_dim / 2] fp32, 16-byte aligned");
  UG_REQUIRE(M < (1 << 30) && N < (1 << 30) && K < (1 << 30), "ug_gemm_bf16_qkv_rope: dims too large");
  const bool fused = head_dim == 128 && N % PBN == 0 && rope_cols % PBN == 0 && K % PBK == 0 && ldq % 8 == 0 && ldx % 8 == 0 && ldw % 8 == 0 &&
                     ldx >= K && ldw >= K && ldq >= N && ug_aligned16(x) && ug_aligned16(w) && ug_aligned16(qkv) &&
                     (!bias || (reinterpret_cast<uintptr_t>(bias) & 7) == 0) && N / PBN <= 64;
  static const bool fused_on = !(getenv("UNIGEN_FUSED_ROPE") && atoi(getenv("UNIGEN_FUSED_ROPE")) == 0);      // A/B switch
  if (!fused || !fused_on) {                    // any other shape: the projection and the rotation as two launches (identical values)
    if (int rc = ug_gemm_bf16(h, x, ldx, 0, w, ldw, 0, qkv, ldq, M, N, K, EPI_BF16, bias, nullptr, 0, 0, nullptr, -1, stream)) return rc;
    if (rope_cols == 0) return UG_OK;
    return ug_rope(qkv, cos_tab, sin_tab, M, L, ldq, (int)(rope_cols / head_dim), head_dim, 0, stream);
  }
  GemmArgs a{};
  a.A = (const bf16_t*)x; a.B = (const bf16_t*)w; a.C = qkv; a.bias = (const bf16_t*)bias;
  a.M = (int)M; a.N = (int)N; a.K = (int)K; a.lda = ldx; a.ldb = ldw; a.ldc = ldq;
  a.rope_cos = cos_tab; a.rope_sin = sin_tab; a.rope_L = (int)L; a.rope_cols = (int)rope_cols;
  a.tail_split = 1; a.tail_private = 1; a.wide_epilogue = 1;
  // tile height: rounds(h) x t(h), the rule of launch() (one round: the smallest height that fits)
  static const int heights[] = {128, 160, 192, 208, 224, 256, 288, 320};
  int hb = 320;
  if (int rc = pick_fused_height(heights, M, N / PBN, "ug_gemm_bf16_qkv_rope", &hb)) return rc;
  a.tiles_m = (int)((M + hb - 1) / hb); a.tiles_n = (int)(N / PBN);
  const dim3 grid(a.tiles_m * a.tiles_n), block(512);
  switch (hb) {
    case 320: hipLaunchKernelGGL((gemm_kernel_p10<EPI_ROPE, false, 10, 10>), grid, block, 0, stream, a); break;
    case 288: hipLaunchKernelGGL((gemm_kernel_p10<EPI_ROPE, false, 9, 9>), grid, block, 0, stream, a); break;
    case 256: hipLaunchKernelGGL((gemm_kernel_p10<EPI_ROPE, false, 8, 8>), grid, block, 0, stream, a); break;
    case 224: hipLaunchKernelGGL((gemm_kernel_p10<EPI_ROPE, false, 7, 7>), grid, block, 0, stream, a); break;
    case 208: hipLaunchKernelGGL((gemm_kernel_p10<EPI_ROPE, false, 7, 6>), grid, block, 0, stream, a); break;
    case 192: hipLaunchKernelGGL((gemm_kernel_p10<EPI_ROPE, false, 6, 6>), grid, block, 0, stream, a); break;
    case 160: hipLaunchKernelGGL((gemm_kernel_p10<EPI_ROPE, false, 5, 5>), grid, block, 0, stream, a); break;
    default: hipLaunchKernelGGL((gemm_kernel_p10<EPI_ROPE, false, 4, 4>), grid, block, 0, stream, a); break;
  }
  UG_CHECK_LAUNCH("ug_gemm_bf16_qkv_rope");
  return UG_OK;
}


// The down projection's dgrad with the SwiGLU backward in its epilogue (EPI_SWIGLU_BWD above): dgu[M, 2I] = swiglu_bwd(gu, dy W_down)
// -- the values of ug_gemm_bf16 (B k-major) followed by ug_swiglu_bwd, bit for bit, without d(act) in HBM.  Shapes the kernel does
// not cover (I % 256, K % 32, unaligned rows) are refused with UG_ERR_ARG: the caller runs the two launches (it owns the d(act) buffer).
extern "C" int ug_gemm_bf16_swiglu_bwd(const ug_handle* h, const void* dy, int64_t ld_dy, const void* w_down, int64_t ldw, const void* gu,
                                       int64_t ld_gu, void* dgu, int64_t ld_dgu, int64_t M, int64_t I, int64_t K, hipStream_t stream) {
  (void)h;
  UG_REQUIRE(M > 0 && I > 0 && K > 0 && M < (1 << 30) && I < (1 << 29) && K < (1 << 30), "ug_gemm_bf16_swiglu_bwd: bad problem M=%ld I=%ld K=%ld", (long)M, (long)I, (long)K);
  UG_REQUIRE(I % PBN == 0 && K % PBK == 0 && I / PBN <= (1 << 20), "ug_gemm_bf16_swiglu_bwd: fused form needs I %% 256 == 0 and K %% 32 == 0 (I=%ld K=%ld): run ug_gemm_bf16 + ug_swiglu_bwd", (long)I, (long)K);
  UG_REQUIRE(ld_dy % 8 == 0 && ldw % 8 == 0 && ld_gu % 8 == 0 && ld_dgu % 8 == 0 && ld_dy >= K && ldw >= I && ld_gu >= 2 * I && ld_dgu >= 2 * I,
             "ug_gemm_bf16_swiglu_bwd: row strides must be multiples of 8 elements and cover the rows");
  UG_REQUIRE(ug_aligned16(dy) && ug_aligned16(w_down) && ug_aligned16(gu) && ug_aligned16(dgu), "ug_gemm_bf16_swiglu_bwd: alignment");
  GemmArgs a{};
  a.A = (const bf16_t*)dy; a.B = (const bf16_t*)w_down; a.C = dgu;
  a.M = (int)M; a.N = (int)I; a.K = (int)K; a.lda = ld_dy; a.ldb = ldw; a.ldc = ld_dgu;
  a.sw_gu = (const bf16_t*)gu; a.ld_gu = ld_gu; a.swiglu_I = (int)I;
  a.tail_split = 1; a.tail_private = 1; a.wide_epilogue = 1;
  static const int heights[] = {128, 160, 192, 208, 224, 256, 272, 288, 320};
  int hb = 320;
  if (int rc = pick_fused_height(heights, M, I / PBN, "ug_gemm_bf16_swiglu_bwd", &hb)) return rc;
  a.tiles_m = (int)((M + hb - 1) / hb); a.tiles_n = (int)(I / PBN);
  const dim3 grid(a.tiles_m * a.tiles_n), block(512);
  switch (hb) {
    case 320: hipLaunchKernelGGL((gemm_kernel_p10<EPI_SWIGLU_BWD, true, 10, 10>), grid, block, 0, stream, a); break;
    case 288: hipLaunchKernelGGL((gemm_kernel_p10<EPI_SWIGLU_BWD, true, 9, 9>), grid, block, 0, stream, a); break;
    case 272: hipLaunchKernelGGL((gemm_kernel_p10<EPI_SWIGLU_BWD, true, 9, 8>), grid, block, 0, stream, a); break;
    case 256: hipLaunchKernelGGL((gemm_kernel_p10<EPI_SWIGLU_BWD, true, 8, 8>), grid, block, 0, stream, a); break;
    case 224: hipLaunchKernelGGL((gemm_kernel_p10<EPI_SWIGLU_BWD, true, 7, 7>), grid, block, 0, stream, a); break;
    case 208: hipLaunchKernelGGL((gemm_kernel_p10<EPI_SWIGLU_BWD, true, 7, 6>), grid, block, 0, stream, a); break;
    case 192: hipLaunchKernelGGL((gemm_kernel_p10<EPI_SWIGLU_BWD, true, 6, 6>), grid, block, 0, stream, a); break;
    case 160: hipLaunchKernelGGL((gemm_kernel_p10<EPI_SWIGLU_BWD, true, 5, 5>), grid, block, 0, stream, a); break;
    default: hipLaunchKernelGGL((gemm_kernel_p10<EPI_SWIGLU_BWD, true, 4, 4>), grid, block, 0, stream, a); break;
  }
  UG_CHECK_LAUNCH("ug_gemm_bf16_swiglu_bwd");
  return UG_OK;
}
