// Single-writer decode layer for UniGen.t2i_generate_ar / generate / mmu_generate (reference models/unigen.py:457-521, which
// drives transformers' Qwen2 one token at a time; modeling_qwen2.py:46-48 MLP, :176-234 attention, :236-286 layer).
//
// Round 6 form.  decode.hip's layer leaves a RAW split-K fp32 accumulator behind every projection (fp32 atomics) and the next launch
// pulls it back through the device coherence point; ~4.5 us per launch were that round trip, the atomic drain and the clears.  Here
// every output element has exactly ONE writer: a workgroup owns a few weight rows for the WHOLE contraction, its waves split K
// (one 256-wide k-slab each, activation fragments in registers), the partial 16x16 MFMA tiles meet in LDS, and the finished value
// leaves by a plain store -- bias + RoPE + cache append for q/k/v, residual add for o / down, SwiGLU for gate/up, fp32 logits for
// the head slice.  No atomics, no accumulators, no clears, no statistics buffers; results are bit-reproducible run to run.
// RMSNorm is applied where the reference applies it: operand = bf16(w * (x * rstd)) (modeling_qwen2.py Qwen2RMSNorm.forward), the
// row statistics recomputed by every workgroup from the finished fp32 residual stream (96 KB, L2-resident).
//   weights stream HBM -> LDS by LDS-DMA (nt), 16 rows x 256 k per tile, source-side XOR swizzle, per-wave ring;
//   workgroup -> weight rows so that 256 workgroups split every projection of the 1.5B model exactly (8 / 6 / 70 / 6 / 32 rows).
#include "common.h"
#include "unigen_hip.h"
#include "vmem_asm.h"
#include <stdlib.h>

namespace {

constexpr int DHD = 128;
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __bf16 bf16pair_t __attribute__((ext_vector_type(2)));

enum { PRO_NORM = 0, PRO_BF16 = 1 };
enum { EPI_QKV = 0, EPI_RESID = 1, EPI_SWIGLU = 2, EPI_STORE = 3 };

struct SwArgs {
  const bf16_t* W; int ldw; int K; int R;
  int nunits, upw;                                         // units of work in all / per workgroup (meaning depends on the epilogue)
  const float* h; const float* norm_w; float eps;          // PRO_NORM: fp32 residual stream [R][K], RMSNorm weight
  const float* pend; int ld_pend; float* x_out;            // PEND: h + bf16round(pend) is the residual stream (a split-K producer's raw
                                                           //   accumulator); the first eight workgroups write it to x_out (!= h)
  const bf16_t* xb; int ldx;                               // PRO_BF16: finished bf16 operand [R][ldx]
  float* h_io;                                             // EPI_RESID: h[r][col] += float(bf16(y))
  bf16_t* act; int ld_act; int I;                          // EPI_SWIGLU
  float* out; int ld_out;                                  // EPI_STORE
  const bf16_t* bias; const float* cs; const float* sn; const int* pos_dev;                     // EPI_QKV
  bf16_t* q_out; int ldq; bf16_t* ck; bf16_t* cv; int Hq, Hk, Tmax, max_pos;
  int* pos_inc; int* len_inc;                              // advanced by workgroup 0 when given (the step's last reader of pos is behind us)
  unsigned long long* trace;                                // UG_SW_TRACE
};

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#define UG_SW_STAMP(i) do { if (a.trace && (threadIdx.x & 63) == 0) a.trace[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)   // UG_SW_TRACE
__device__ __forceinline__ float silu_bf(float g) { return bf2f(f2bf(g / (1.f + __expf(-g)))); }

// Workgroup -> weight rows.  A tile = 16 slots (the MFMA's B columns); `units` are what a workgroup owns:
//   EPI_RESID / EPI_STORE: unit = weight row, 16 per tile.
//   EPI_SWIGLU: unit = hidden unit c, 8 per tile: slots 0-7 = gate rows c, slots 8-15 = up rows I + c of the same eight units (gate and
//               up of a unit meet in one wave after the reduction).
//   EPI_QKV: unit = four rotary pairs of one head, ONE per tile: slots 0-3 = dims 4j..4j+3, slots 4-7 = dims 64+4j..64+4j+3 (16 units
//            per head: the rotation's partner is four lanes away), slots 8-15 unused.
template <int EPI>
struct RowMap {
  static constexpr int UPT = EPI == EPI_SWIGLU ? 8 : EPI == EPI_QKV ? 1 : 16;           // units per tile
  static __device__ __forceinline__ int rel_row(const SwArgs& a, int s) {                // weight row of slot s relative to the tile's first
    if constexpr (EPI == EPI_SWIGLU) return s < 8 ? s : a.I + (s - 8);
    else if constexpr (EPI == EPI_QKV) return ((s & 7) >> 2) * 64 + (s & 3);
    else return s;
  }
  static __device__ __forceinline__ int row0(int unit) {                                 // first weight row of the tile that starts at `unit`
    if constexpr (EPI == EPI_QKV) return (unit >> 4) * DHD + (unit & 15) * 4;
    else return unit;
  }
  static __device__ __forceinline__ int nvalid(int u0, int u1, int t) { return min(max(u1 - (u0 + UPT * t), 0), UPT); }
  static __device__ __forceinline__ bool slot_valid(int s, int nv) {
    if constexpr (EPI == EPI_SWIGLU) return (s & 7) < nv;
    else if constexpr (EPI == EPI_QKV) return nv > 0 && s < 8;
    else return s < nv;
  }
};

// NW waves per workgroup; wave w owns k-slabs w, w + NW, ... (NS of them) and walks MAXT tiles of 16 weight rows per slab
// (items j * MAXT + t) through a RING-slot LDS ring of MAXI KiB slots.  PRO_NORM needs K == 256 NW and NS == 1.
// PRE = ring slots staged BEFORE the operand prologue.  A CU issues about one 1 KB vector-memory instruction per 14 ns, and a wave
// reaches its operand wait only after its own DMA burst: with the whole ring up front the prologue's workgroup barrier completed 5-7 us
// into the gate/up launch (tools/probes/decode_sw_probe.cpp, PROBE_TRACE).  One slot per wave (48 KB per CU) keeps HBM busy meanwhile.
template <int PRO, int EPI, int NW, int MAXT, int NS, int RING, int MAXI, bool PEND = false, int PRE_ = 0, bool DEFER = false>
__global__ __launch_bounds__(64 * NW) void gemv_sw_kernel(SwArgs a) {
  constexpr int NITEMS = MAXT * NS;
  constexpr int PRE = PRE_ ? PRE_ : (PRO == PRO_NORM ? 1 : RING);
  static_assert(PRE >= 1 && PRE <= RING && (PRO == PRO_NORM || PRE == RING) && (!PEND || PRO == PRO_NORM), "staging split");
  static_assert(RING <= NITEMS && MAXT <= RING * MAXI, "ring too small for the reduction image");
  static_assert(PRO != PRO_NORM || NS == 1, "PRO_NORM: one slab per wave");
  static_assert(MAXT <= NW, "wave t finishes tile t");
  __shared__ __attribute__((aligned(1024))) char tile[NW][RING][MAXI * 1024];
  __shared__ __attribute__((aligned(16))) float wn[PRO == PRO_NORM ? NW : 1][256];
  __shared__ float ssp[PRO == PRO_NORM ? NW : 1][16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, slot = lane & 15;
  const int u0 = blockIdx.x * a.upw, u1 = min(a.nunits, u0 + a.upw);
  if (u0 >= u1) return;
  const int K = a.K, nslabs = K >> 8;
  const int arow = min(slot, a.R - 1);                       // the MFMA's A row of this lane (rows past R repeat the last one)

  UG_SW_STAMP(0);                                                       // UG_SW_TRACE
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  typedef RowMap<EPI> RM;

  // ---- what the epilogue needs from memory is requested before anything else: the finishing wave of tile t (wave t) would otherwise
  // start a dependent round trip (residual values; position -> RoPE table) when everything else is done
  float hold[4] = {0.f, 0.f, 0.f, 0.f};
  int pos0 = 0;
  if constexpr (EPI == EPI_RESID) {
    if (wave < MAXT) {
      const int col = min(u0 + 16 * wave + slot, u1 - 1);
#pragma unroll
      for (int j = 0; j < 4; ++j) hold[j] = a.h_io[__umul24(min(g * 4 + j, a.R - 1), a.nunits) + col];
    }
  }
  if constexpr (EPI == EPI_QKV) pos0 = *a.pos_dev;

  // ---- operand loads (the critical chain), then the weight DMA
  f32x4_t xa[PRO == PRO_NORM ? 8 : 1][2], pa[PEND ? 8 : 1][2];
  f32x4_t wv;
  bf16x8_t xf[NS][8];
  if constexpr (PRO == PRO_NORM) {
    const uint32_t xo = (uint32_t)(__umul24(arow, K) + wave * 256 + g * 8) * 4u;
#define UG_LDX(u) ld16<(u) * 128>(xa[u][0], (uint64_t)a.h, xo); ld16<(u) * 128 + 16>(xa[u][1], (uint64_t)a.h, xo);
    UG_LDX(0) UG_LDX(1) UG_LDX(2) UG_LDX(3) UG_LDX(4) UG_LDX(5) UG_LDX(6) UG_LDX(7)
#undef UG_LDX
    if constexpr (PEND) {
      const uint32_t po = (uint32_t)(__umul24(arow, a.ld_pend) + wave * 256 + g * 8) * 4u;
#define UG_LDX(u) ld16<(u) * 128>(pa[u][0], (uint64_t)a.pend, po); ld16<(u) * 128 + 16>(pa[u][1], (uint64_t)a.pend, po);
      UG_LDX(0) UG_LDX(1) UG_LDX(2) UG_LDX(3) UG_LDX(4) UG_LDX(5) UG_LDX(6) UG_LDX(7)
#undef UG_LDX
    }
    ld16<0>(wv, (uint64_t)a.norm_w, (uint32_t)(wave * 256 + lane * 4) * 4u);
  } else {
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      const int slab = min(wave + j * NW, nslabs - 1);
      const uint32_t xo = (uint32_t)(__umul24(arow, a.ldx) + slab * 256 + g * 8) * 2u;
#define UG_LDX(u) ld16<(u) * 64>(xf[j][u], (uint64_t)a.xb, xo);
      UG_LDX(0) UG_LDX(1) UG_LDX(2) UG_LDX(3) UG_LDX(4) UG_LDX(5) UG_LDX(6) UG_LDX(7)
#undef UG_LDX
    }
  }
  // byte offset of this lane inside a tile for DMA instruction i (slots 2i, 2i + 1 x 512 bytes; the bank swizzle -- 16-byte chunk
  // index ^ slot -- is applied on the SOURCE side, LDS-DMA writes lane-linear): the same for every tile of the workgroup
  uint32_t voff[MAXI];
#pragma unroll
  for (int i = 0; i < MAXI; ++i) {
    const int sl = 2 * i + (lane >> 5);
    voff[i] = (uint32_t)(RM::rel_row(a, sl) * a.ldw + (((lane & 31) ^ sl) << 3)) * 2u;
  }
  // stage item k (tile t = k % MAXT of slab j = k / MAXT) into ring slot k % RING.  ALWAYS MAXI instructions: the waits below count
  // instructions.  A slot without a weight row reads the tile's first 16 bytes (one line for the whole wave) into LDS nobody uses.
  auto stage = [&](int k) {
    const int t = k % MAXT, j = k / MAXT;
    const int slab = wave_u + j * NW;
    const int nv = slab < nslabs ? RM::nvalid(u0, u1, t) : 0;
    const int r0 = RM::row0(nv > 0 ? u0 + RM::UPT * t : u0);
    const uint64_t base = (uint64_t)a.W + ((int64_t)r0 * a.ldw + min(slab, nslabs - 1) * 256) * 2;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_addr_of(tile[wave][k % RING]));
#pragma unroll
    for (int i = 0; i < MAXI; ++i)       // aux nt: every weight byte is read once per step by one CU (guide, price list row nt-weights)
      dma16_nt(base, RM::slot_valid(2 * i + (lane >> 5), nv) ? voff[i] : 0u, dst + i * 1024);
  };
  // DMA instructions issued after item k's own when its fragments are read: items k + 1 .. min(NITEMS, k + RING) - 1
#define UG_SW_BEHIND(k) ((((k) + RING < NITEMS ? (k) + RING : NITEMS) - 1 - (k)) * MAXI)
#pragma unroll
  for (int k = 0; k < PRE; ++k) stage(k);
  UG_SW_STAMP(1);                                                       // UG_SW_TRACE

  if constexpr (PRO == PRO_NORM) {
    wait_vm<PRE * MAXI>();                                     // the operand loads are older than every DMA instruction
#pragma unroll
    for (int u = 0; u < 8; ++u) { tie(xa[u][0]); tie(xa[u][1]); }
    tie(wv);
    if constexpr (PEND) {
      // residual add of the split-K producer in front of us: the Linear's bf16 output joins the fp32 stream (Qwen2DecoderLayer.forward)
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        tie(pa[u][0]); tie(pa[u][1]);
#pragma unroll
        for (int e = 0; e < 4; ++e) { xa[u][0][e] += bf2f(f2bf(pa[u][0][e])); xa[u][1][e] += bf2f(f2bf(pa[u][1][e])); }
        if (a.x_out && blockIdx.x == u && slot < a.R) {                    // eight workgroups x six waves x one k-step cover the whole stream
          float* xo = a.x_out + (__umul24(slot, K) + wave * 256 + u * 32 + g * 8);
          *reinterpret_cast<f32x4_t*>(xo) = xa[u][0];
          *reinterpret_cast<f32x4_t*>(xo + 4) = xa[u][1];
        }
      }
    }
    UG_SW_STAMP(2);                                                       // UG_SW_TRACE
    *reinterpret_cast<f32x4_t*>(&wn[wave][lane * 4]) = wv;
    float ss = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      ss += xa[u][0][0] * xa[u][0][0] + xa[u][0][1] * xa[u][0][1] + xa[u][0][2] * xa[u][0][2] + xa[u][0][3] * xa[u][0][3];
      ss += xa[u][1][0] * xa[u][1][0] + xa[u][1][1] * xa[u][1][1] + xa[u][1][2] * xa[u][1][2] + xa[u][1][3] * xa[u][1][3];
    }
    ss += __shfl_xor(ss, 16, 64);
    ss += __shfl_xor(ss, 32, 64);
    if (g == 0) ssp[wave][slot] = ss;
    float rs = 1.f;
    if constexpr (!DEFER) {
      lds_barrier();
      float tot = 0.f;
#pragma unroll
      for (int ww = 0; ww < NW; ++ww) tot += ssp[ww][slot];    // same order in every workgroup: one value of rstd per row everywhere
      rs = rsqrtf(tot / (float)K + a.eps);
    } else {
      // DEFER: the row's rstd is a scalar of the output row, so it can multiply the finished contraction instead of the operand
      // (decode.hip's convention: bf16(w x) instead of bf16(w x rstd), the same value up to one bf16 rounding of the operand).  No
      // workgroup barrier ahead of the first MFMA then: a wave whose loads were served late no longer holds the other five back.
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (this wave's own copy of the norm weights)
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float4 w0 = *reinterpret_cast<const float4*>(&wn[wave][u * 32 + g * 8]);
      const float4 w1 = *reinterpret_cast<const float4*>(&wn[wave][u * 32 + g * 8 + 4]);
      // Qwen2RMSNorm.forward: weight * (x * rsqrt(mean(x^2) + eps)), then the Linear's autocast rounds its input to bf16
      const uint32_t p0 = pack_bf2(w0.x * (xa[u][0][0] * rs), w0.y * (xa[u][0][1] * rs));
      const uint32_t p1 = pack_bf2(w0.z * (xa[u][0][2] * rs), w0.w * (xa[u][0][3] * rs));
      const uint32_t p2 = pack_bf2(w1.x * (xa[u][1][0] * rs), w1.y * (xa[u][1][1] * rs));
      const uint32_t p3 = pack_bf2(w1.z * (xa[u][1][2] * rs), w1.w * (xa[u][1][3] * rs));
      xf[0][u] = __builtin_bit_cast(bf16x8_t, make_uint4(p0, p1, p2, p3));
    }
#pragma unroll
    for (int k = PRE; k < RING; ++k) stage(k);
  }

  if constexpr (PRO == PRO_BF16) {
    wait_vm<RING * MAXI>();
#pragma unroll
    for (int j = 0; j < NS; ++j)
#pragma unroll
      for (int u = 0; u < 8; ++u) tie(xf[j][u]);
  }
  UG_SW_STAMP(3);                                                       // UG_SW_TRACE
  float rope_c = 1.f, rope_s = 0.f, bias_v = 0.f;
  if constexpr (EPI == EPI_QKV) {
    if (wave < MAXT && u0 + wave < u1 && slot < 8) {
      const int unit = u0 + wave, w = slot & 7;
      const int d = (w >> 2) * 64 + (unit & 15) * 4 + (w & 3), head = unit >> 4;
      const int pos = min(pos0, a.max_pos - 1);
      if (a.bias) bias_v = bf2f(a.bias[head * DHD + d]);
      if (head < a.Hq + a.Hk) { rope_c = a.cs[(int64_t)pos * (DHD / 2) + (d & 63)]; rope_s = a.sn[(int64_t)pos * (DHD / 2) + (d & 63)]; }
    }
  }
  f32x4_t acc[MAXT];
#pragma unroll
  for (int t = 0; t < MAXT; ++t) acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < NITEMS; ++k) {
    const int t = k % MAXT, j = k / MAXT;
    wait_vm_n(UG_SW_BEHIND(k));
    if (k == 0) { UG_SW_STAMP(4); }                                       // UG_SW_TRACE
    const char* tr = tile[wave][k % RING] + slot * 512;
    bf16x8_t wf[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) wf[u] = *reinterpret_cast<const bf16x8_t*>(tr + (((u * 4 + g) ^ slot) << 4));
    if (k + RING < NITEMS) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this slot's fragments are in registers before the DMA overwrites it
      stage(k + RING);
    }
    if (wave + j * NW < nslabs) {                            // (a wave's last slab may not exist: its tile holds filler)
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[j][u], wf[u], acc[t], 0, 0, 0);
    }
  }
  UG_SW_STAMP(5);                                                       // UG_SW_TRACE
  // ---- the NW partial tiles meet in LDS (each wave parks its own in its OWN ring area: nobody else reads that), fixed order
  {
    f32x4_t* red = reinterpret_cast<f32x4_t*>(&tile[wave][0][0]);
#pragma unroll
    for (int t = 0; t < MAXT; ++t) red[t * 64 + lane] = acc[t];
  }
  lds_barrier();
  UG_SW_STAMP(6);                                                       // UG_SW_TRACE
  if (a.pos_inc && blockIdx.x == 0 && threadIdx.x == 0) { ++*a.pos_inc; ++*a.len_inc; }
#pragma unroll 1
  for (int t = wave; t < MAXT; t += NW) {
    if (RM::nvalid(u0, u1, t) == 0) break;
    f32x4_t v = reinterpret_cast<const f32x4_t*>(&tile[0][0][0])[t * 64 + lane];
#pragma unroll
    for (int ww = 1; ww < NW; ++ww) v += reinterpret_cast<const f32x4_t*>(&tile[ww][0][0])[t * 64 + lane];
    if constexpr (PRO == PRO_NORM && DEFER) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float tot = 0.f;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) tot += ssp[ww][g * 4 + j];
        v[j] *= rsqrtf(tot / (float)K + a.eps);
      }
    }
    if constexpr (EPI == EPI_SWIGLU) {
      // act = bf16( bf16(silu(bf16 gate)) * bf16 up )  (Qwen2MLP.forward under bf16 autocast)
      f32x4_t up;
#pragma unroll
      for (int j = 0; j < 4; ++j) up[j] = __shfl_down(v[j], 8, 64);
      const int unit = u0 + 8 * t + slot;
      if (slot < 8 && unit < u1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = g * 4 + j;
          if (r < a.R) a.act[__umul24(r, a.ld_act) + unit] = f2bf(silu_bf(bf2f(f2bf(v[j]))) * bf2f(f2bf(up[j])));
        }
      }
    } else if constexpr (EPI == EPI_RESID) {
      const int col = u0 + 16 * t + slot;
      if (col < u1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = g * 4 + j;
          if (r < a.R) a.h_io[__umul24(r, a.nunits) + col] = hold[j] + bf2f(f2bf(v[j]));
        }
      }
    } else if constexpr (EPI == EPI_STORE) {
      const int col = u0 + 16 * t + slot;
      if (col < u1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = g * 4 + j;
          if (r < a.R) a.out[(int64_t)r * a.ld_out + col] = v[j];
        }
      }
    } else {
#pragma clang fp contract(off)
      // bias -> bf16 (the Linear's output), rotate-half RoPE with separately rounded products (modeling_qwen2.py apply_rotary_pos_emb
      // on bf16 tensors; same arithmetic as rope_at_kernel / finish_qkv_tile), q -> q_out, k / v -> cache[pos]
      const int unit = u0 + t;
      const int w = slot & 7, hi = w >> 2;
      const int head = unit >> 4, d = hi * 64 + (unit & 15) * 4 + (w & 3);
      const int col = head * DHD + d;
      const bool valid = unit < u1 && slot < 8;
      const float bv = bias_v, c = rope_c, s = rope_s;      // (requested at the kernel's start: this wave finishes tile t == wave)
      const bool rot = head < a.Hq + a.Hk;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = g * 4 + j;
        const float x = bf2f(f2bf(v[j] + bv));
        const float xp = __shfl_xor(x, 4, 64);                 // the rotary partner: same unit, other half
        float y = x;
        if (rot) {
          const float p = x * c, q = xp * s;
          y = bf2f(f2bf(hi ? p + q : p - q));
        }
        if (valid && r < a.R) {
          if (head < a.Hq) a.q_out[__umul24(r, a.ldq) + col] = f2bf(y);
          else if (pos0 < a.Tmax) {
            const bool is_k = head < a.Hq + a.Hk;
            const int hk = is_k ? head - a.Hq : head - a.Hq - a.Hk;
            bf16_t* dst = (is_k ? a.ck : a.cv) + (((int64_t)r * a.Hk + hk) * a.Tmax + pos0) * DHD;
            dst[d] = f2bf(y);
          }
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // UG_SW_TRACE
  UG_SW_STAMP(7);                                                       // UG_SW_TRACE
}

// ------------------------------------------------------------------ cache attention with a finished q
// One workgroup (8 waves) per (row, query head); keys [0, pos] -- the new token's k / v were appended by the q/k/v projection's
// epilogue.  Same chunking / LDS-DMA K tiles / dot2 arithmetic as decode.hip's attn_decode_fused_kernel, without its prologue.
constexpr int ADQ_WAVES = 8;
__global__ __launch_bounds__(64 * ADQ_WAVES) void attn_decode_q_kernel(const bf16_t* __restrict__ q, int ldq, const bf16_t* __restrict__ ck,
                                                                       const bf16_t* __restrict__ cv, const uint8_t* __restrict__ key_valid,
                                                                       bf16_t* __restrict__ o, int ldo, int R, int HKV, int Tmax,
                                                                       const int* __restrict__ pos_dev, float scale) {
  __shared__ __attribute__((aligned(16))) bf16_t qb[DHD];
  __shared__ float om[ADQ_WAVES][DHD];
  __shared__ float ml[ADQ_WAVES][2];
  __shared__ __attribute__((aligned(1024))) char ktile[ADQ_WAVES][64 * DHD * 2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int per = gridDim.y, hq = blockIdx.y;
  const int grp = blockIdx.x + 8 * blockIdx.z;            // (row, kv head) group: its query heads share one XCD's L2
  if (grp >= R * HKV) return;
  int r, hk;
  if ((HKV & (HKV - 1)) == 0) { r = grp >> (31 - __builtin_clz(HKV)); hk = grp & (HKV - 1); }
  else { r = grp / HKV; hk = grp - r * HKV; }
  const int h = hk * per + hq;
  const bf16_t* kb = ck + ((int64_t)r * HKV + hk) * Tmax * DHD;
  const bf16_t* vb = cv + ((int64_t)r * HKV + hk) * Tmax * DHD;
  const int kq = lane >> 4, dc = lane & 15;
  bf16x8_t vf[16];
  auto load_chunk = [&](int t0, int last) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int k = i * 4 + kq;
      const bf16_t* src = kb + (int64_t)min(t0 + k, last) * DHD + ((dc ^ (k & 15)) << 3);
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(ktile[wave] + i * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
      const int tt = min(t0 + jj * 4 + kq, last);
      vf[jj] = *reinterpret_cast<const bf16x8_t*>(vb + (int64_t)tt * DHD + dc * 8);
    }
  };
  uint4 qv = make_uint4(0, 0, 0, 0);
  if (threadIdx.x < DHD / 8) qv = *reinterpret_cast<const uint4*>(q + (__umul24(r, ldq) + h * DHD + threadIdx.x * 8));
  const int len = min(*pos_dev + 1, Tmax);
  if (wave * 64 < len) load_chunk(wave * 64, len - 1);
  if (threadIdx.x < DHD / 8) *reinterpret_cast<uint4*>(&qb[threadIdx.x * 8]) = qv;
  lds_barrier();
  float m = -INFINITY, l = 0.f;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll 1
  for (int t0 = wave * 64; t0 < len; t0 += 64 * ADQ_WAVES) {
    const int t = t0 + lane;
    float s = -INFINITY;
    if (t0 != wave * 64) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      load_chunk(t0, len - 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (t < len && (!key_valid || key_valid[(int64_t)r * Tmax + t])) {
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
    const float pb = bf2f(f2bf(p));                      // P is rounded to bf16 before P.V like the bf16 SDPA paths
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] *= alpha;
#pragma unroll
    for (int jj = 0; jj < 16; jj += 2) {
      const uint32_t pp = pack_bf2(__shfl(pb, jj * 4 + kq, 64), __shfl(pb, (jj + 1) * 4 + kq, 64));
      const uint4 va = __builtin_bit_cast(uint4, vf[jj]), vb2 = __builtin_bit_cast(uint4, vf[jj + 1]);
      const uint32_t wa[4] = {va.x, va.y, va.z, va.w}, wb[4] = {vb2.x, vb2.y, vb2.z, vb2.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint32_t lo = __builtin_amdgcn_perm(wb[i], wa[i], 0x05040100u);
        const uint32_t hi = __builtin_amdgcn_perm(wb[i], wa[i], 0x07060302u);
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
  lds_barrier();
  if (threadIdx.x < DHD) {
    float M = ml[0][0];
#pragma unroll
    for (int w = 1; w < ADQ_WAVES; ++w) M = fmaxf(M, ml[w][0]);
    float L = 0.f, O = 0.f;
#pragma unroll
    for (int w = 0; w < ADQ_WAVES; ++w) {
      const float wgt = (ml[w][0] == -INFINITY) ? 0.f : __expf(ml[w][0] - M);
      L += wgt * ml[w][1];
      O += wgt * om[w][threadIdx.x];
    }
    const float inv = L > 0.f ? 1.f / L : 0.f;
    o[__umul24(r, ldo) + h * DHD + threadIdx.x] = f2bf(O * inv);
  }
}

unsigned long long* g_trace = nullptr;                                  // UG_SW_TRACE
int cu_count() {
  static const int n = [] {
    const char* e = getenv("UNIGEN_DECODE_SW_WGS");
    if (e && atoi(e) > 0) return atoi(e);
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess || p.multiProcessorCount <= 0) return 256;
    return p.multiProcessorCount;
  }();
  return n;
}

// units per workgroup: one workgroup per CU when the tile budget allows, else as many units as MAXT tiles hold
int units_per_wg(int nunits, int max_upw) {
  const int ncu = cu_count();
  int upw = (nunits + ncu - 1) / ncu;
  if (upw < 1) upw = 1;
  return upw > max_upw ? max_upw : upw;
}

#define UG_SW_COMMON(name)                                                                                                                   \
  UG_REQUIRE(R > 0 && R <= 16 && W && ldw % 8 == 0 && ug_aligned16(W) && N > 0 && (int64_t)N * ldw < (1ll << 31) && ldw < (1 << 24),        \
             name ": need 1 <= rows <= 16, 16-byte aligned weight rows, N * ldw < 2^31 (rows=%ld N=%ld ldw=%ld)", (long)R, (long)N, (long)ldw)

}  // namespace

extern "C" void ug_decode_sw_set_trace(unsigned long long* p) { g_trace = p; }      // UG_SW_TRACE
extern "C" int ug_decode_sw_supported(int64_t hidden, int64_t inter, int64_t q_dim, int head_dim) {
  return hidden == 1536 && q_dim % 256 == 0 && q_dim / 256 <= 6 && inter % 256 == 0 && inter / 256 <= 36 && head_dim == DHD ? 1 : 0;
}

#define UG_SW_PEND_ARGS(name)                                                                                                         \
  UG_REQUIRE(pend == nullptr || (ld_pend >= H && ld_pend % 4 == 0 && ld_pend < (1 << 20) && ug_aligned16(pend) && x_out != h &&           \
                                 (x_out == nullptr || ug_aligned16(x_out))),                                                           \
             name ": pending accumulator must be 16-byte aligned fp32 rows and x_out a buffer other than h");                         \
  UG_REQUIRE(pend != nullptr || x_out == nullptr, name ": x_out is written only together with a pending accumulator");               \
  a.pend = pend; a.ld_pend = (int)ld_pend; a.x_out = x_out

extern "C" int ug_decode_sw_qkv(const float* h, const float* pend, int64_t ld_pend, float* x_out, const float* norm_w, float eps,
                                int64_t R, int64_t H, const void* W, int64_t ldw,
                                const void* bias, const float* cos_tab, const float* sin_tab, const int* pos_dev, void* q_out,
                                int64_t ldq, void* cache_k, void* cache_v, int Hq, int Hkv, int head_dim, int64_t Tmax,
                                int64_t max_pos, hipStream_t st) {
  const int64_t N = (int64_t)(Hq + 2 * Hkv) * DHD;
  UG_SW_COMMON("ug_decode_sw_qkv");
  UG_REQUIRE(h && norm_w && cos_tab && sin_tab && pos_dev && q_out && cache_k && cache_v && head_dim == DHD && H == 1536 && ldw >= H &&
                 ug_aligned16(h) && ug_aligned16(norm_w) && Tmax > 0 && max_pos > 0 && ldq >= (int64_t)Hq * DHD && ldq < (1 << 20),
             "ug_decode_sw_qkv: bad args (hidden must be 1536, head_dim 128; hidden=%ld head_dim=%d)", (long)H, head_dim);
  SwArgs a{};
  a.W = (const bf16_t*)W; a.ldw = (int)ldw; a.K = (int)H; a.R = (int)R;
  a.nunits = (Hq + 2 * Hkv) * 16; a.upw = units_per_wg(a.nunits, 2);      // (one unit per tile, two tiles)
  a.h = h; a.norm_w = norm_w; a.eps = eps;
  a.bias = (const bf16_t*)bias; a.cs = cos_tab; a.sn = sin_tab; a.pos_dev = pos_dev; a.q_out = (bf16_t*)q_out; a.ldq = (int)ldq;
  a.ck = (bf16_t*)cache_k; a.cv = (bf16_t*)cache_v; a.Hq = Hq; a.Hk = Hkv; a.Tmax = (int)Tmax; a.max_pos = (int)max_pos;
  UG_SW_PEND_ARGS("ug_decode_sw_qkv");
  const unsigned grid = (unsigned)((a.nunits + a.upw - 1) / a.upw);
  UG_REQUIRE(!pend || grid >= 8, "ug_decode_sw_qkv: fewer than eight workgroups cannot write x_out");
  a.trace = g_trace;                                                    // UG_SW_TRACE
  if (pend) hipLaunchKernelGGL((gemv_sw_kernel<PRO_NORM, EPI_QKV, 6, 2, 1, 2, 4, true>), dim3(grid), dim3(64 * 6), 0, st, a);
  else hipLaunchKernelGGL((gemv_sw_kernel<PRO_NORM, EPI_QKV, 6, 2, 1, 2, 4>), dim3(grid), dim3(64 * 6), 0, st, a);
  UG_CHECK_LAUNCH("ug_decode_sw_qkv");
  return UG_OK;
}

extern "C" int ug_decode_sw_gate_up(const float* h, const float* pend, int64_t ld_pend, float* x_out, const float* norm_w, float eps,
                                    int64_t R, int64_t H, const void* W, int64_t ldw,
                                    int64_t I, void* act, int64_t ld_act, hipStream_t st) {
  const int64_t N = 2 * I;
  UG_SW_COMMON("ug_decode_sw_gate_up");
  UG_REQUIRE(h && norm_w && act && H == 1536 && ldw >= H && I > 0 && ld_act >= I && ld_act < (1 << 20) && ug_aligned16(h) && ug_aligned16(norm_w),
             "ug_decode_sw_gate_up: bad args (hidden must be 1536; hidden=%ld)", (long)H);
  SwArgs a{};
  a.W = (const bf16_t*)W; a.ldw = (int)ldw; a.K = (int)H; a.R = (int)R;
  a.nunits = (int)I; a.upw = units_per_wg(a.nunits, 40);
  a.h = h; a.norm_w = norm_w; a.eps = eps;
  a.act = (bf16_t*)act; a.ld_act = (int)ld_act; a.I = (int)I;
  UG_SW_PEND_ARGS("ug_decode_sw_gate_up");
  const unsigned grid = (unsigned)((a.nunits + a.upw - 1) / a.upw);
  UG_REQUIRE(!pend || grid >= 8, "ug_decode_sw_gate_up: fewer than eight workgroups cannot write x_out");
  a.trace = g_trace;                                                    // UG_SW_TRACE
  static const int variant = [] { const char* e = getenv("UNIGEN_SW_VARIANT"); return e ? atoi(e) : 0; }();            // UG_SW_TRACE
  if (!pend && variant == 1) hipLaunchKernelGGL((gemv_sw_kernel<PRO_NORM, EPI_SWIGLU, 6, 5, 1, 3, 8, false, 3>), dim3(grid), dim3(64 * 6), 0, st, a);         // UG_SW_TRACE
  else if (!pend && variant == 2) hipLaunchKernelGGL((gemv_sw_kernel<PRO_NORM, EPI_SWIGLU, 6, 5, 1, 3, 8, false, 1, true>), dim3(grid), dim3(64 * 6), 0, st, a);   // UG_SW_TRACE
  else if (!pend && variant == 3) hipLaunchKernelGGL((gemv_sw_kernel<PRO_NORM, EPI_SWIGLU, 6, 5, 1, 3, 8, false, 3, true>), dim3(grid), dim3(64 * 6), 0, st, a);   // UG_SW_TRACE
  else if (!pend && variant == 4) hipLaunchKernelGGL((gemv_sw_kernel<PRO_NORM, EPI_SWIGLU, 6, 5, 1, 3, 8, false, 2, true>), dim3(grid), dim3(64 * 6), 0, st, a);   // UG_SW_TRACE
  else                                                                                                                   // UG_SW_TRACE
  if (pend) hipLaunchKernelGGL((gemv_sw_kernel<PRO_NORM, EPI_SWIGLU, 6, 5, 1, 3, 8, true>), dim3(grid), dim3(64 * 6), 0, st, a);
  else hipLaunchKernelGGL((gemv_sw_kernel<PRO_NORM, EPI_SWIGLU, 6, 5, 1, 3, 8>), dim3(grid), dim3(64 * 6), 0, st, a);
  UG_CHECK_LAUNCH("ug_decode_sw_gate_up");
  return UG_OK;
}

extern "C" int ug_decode_sw_resid(const void* x, int64_t ldx, int64_t R, const void* W, int64_t ldw, int64_t N, int64_t K, float* h,
                                  hipStream_t st) {
  UG_SW_COMMON("ug_decode_sw_resid");
  UG_REQUIRE(x && h && K > 0 && K % 256 == 0 && ldx >= K && ldx % 8 == 0 && ldx < (1 << 20) && ldw >= K && ug_aligned16(x) && N < (1 << 20),
             "ug_decode_sw_resid: bad args (K %% 256 == 0 required; K=%ld)", (long)K);
  SwArgs a{};
  a.W = (const bf16_t*)W; a.ldw = (int)ldw; a.K = (int)K; a.R = (int)R;
  a.nunits = (int)N; a.xb = (const bf16_t*)x; a.ldx = (int)ldx; a.h_io = h;
  const int nslabs = (int)(K / 256);
  if (nslabs <= 6) {
    a.upw = units_per_wg(a.nunits, 8);
    const unsigned grid = (unsigned)((a.nunits + a.upw - 1) / a.upw);
  a.trace = g_trace;                                                    // UG_SW_TRACE
    hipLaunchKernelGGL((gemv_sw_kernel<PRO_BF16, EPI_RESID, 6, 1, 1, 1, 4>), dim3(grid), dim3(64 * 6), 0, st, a);
  } else {
    UG_REQUIRE(nslabs <= 36, "ug_decode_sw_resid: K = %ld beyond the 36 k-slabs of this build", (long)K);
    a.upw = units_per_wg(a.nunits, 8);
    const unsigned grid = (unsigned)((a.nunits + a.upw - 1) / a.upw);
  a.trace = g_trace;                                                    // UG_SW_TRACE
    hipLaunchKernelGGL((gemv_sw_kernel<PRO_BF16, EPI_RESID, 12, 1, 3, 3, 4>), dim3(grid), dim3(64 * 12), 0, st, a);
  }
  UG_CHECK_LAUNCH("ug_decode_sw_resid");
  return UG_OK;
}

extern "C" int ug_decode_sw_head(const float* h, const float* pend, int64_t ld_pend, float* x_out, const float* norm_w, float eps,
                                 int64_t R, int64_t H, const void* W, int64_t ldw,
                                 int64_t N, float* logits, int64_t ld_logits, int* pos_inc, int* len_inc, hipStream_t st) {
  UG_SW_COMMON("ug_decode_sw_head");
  UG_REQUIRE((pos_inc == nullptr) == (len_inc == nullptr), "ug_decode_sw_head: pos_inc / len_inc come together");
  UG_REQUIRE(h && norm_w && logits && H == 1536 && ldw >= H && ld_logits >= N && ug_aligned16(h) && ug_aligned16(norm_w),
             "ug_decode_sw_head: bad args (hidden must be 1536; hidden=%ld)", (long)H);
  SwArgs a{};
  a.W = (const bf16_t*)W; a.ldw = (int)ldw; a.K = (int)H; a.R = (int)R;
  a.nunits = (int)N; a.upw = units_per_wg(a.nunits, 32);
  a.h = h; a.norm_w = norm_w; a.eps = eps; a.out = logits; a.ld_out = (int)ld_logits; a.pos_inc = pos_inc; a.len_inc = len_inc;
  UG_SW_PEND_ARGS("ug_decode_sw_head");
  const unsigned grid = (unsigned)((a.nunits + a.upw - 1) / a.upw);
  UG_REQUIRE(!pend || !x_out || grid >= 8, "ug_decode_sw_head: fewer than eight workgroups cannot write x_out");
  a.trace = g_trace;                                                    // UG_SW_TRACE
  if (pend) hipLaunchKernelGGL((gemv_sw_kernel<PRO_NORM, EPI_STORE, 6, 2, 1, 2, 8, true>), dim3(grid), dim3(64 * 6), 0, st, a);
  else hipLaunchKernelGGL((gemv_sw_kernel<PRO_NORM, EPI_STORE, 6, 2, 1, 2, 8>), dim3(grid), dim3(64 * 6), 0, st, a);
  UG_CHECK_LAUNCH("ug_decode_sw_head");
  return UG_OK;
}

extern "C" int ug_attn_decode_q(const void* q, int64_t ldq, const void* cache_k, const void* cache_v, const uint8_t* key_valid, void* o,
                                int64_t ldo, int64_t rows, int H, int HKV, int head_dim, int64_t Tmax, const int* pos_dev, float scale,
                                hipStream_t st) {
  UG_REQUIRE(rows > 0 && head_dim == DHD && HKV > 0 && H % HKV == 0 && q && cache_k && cache_v && o && pos_dev && ug_aligned16(q) &&
                 ldq % 8 == 0 && ldq < (1 << 20) && ldo < (1 << 20) && Tmax > 0,
             "ug_attn_decode_q: bad args");
  const dim3 grid(8u, (unsigned)(H / HKV), (unsigned)((rows * HKV + 7) / 8));
  hipLaunchKernelGGL(attn_decode_q_kernel, grid, dim3(64 * ADQ_WAVES), 0, st, (const bf16_t*)q, (int)ldq, (const bf16_t*)cache_k,
                     (const bf16_t*)cache_v, key_valid, (bf16_t*)o, (int)ldo, (int)rows, HKV, (int)Tmax, pos_dev, scale);
  UG_CHECK_LAUNCH("ug_attn_decode_q");
  return UG_OK;
}
