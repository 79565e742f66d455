// Single-writer decode projections for UniGen.t2i_generate_ar / generate / mmu_generate (reference models/unigen.py:457-521, which
// drives transformers' Qwen2 one token at a time; modeling_qwen2.py:46-48 MLP, :236-286 layer, Qwen2Model.norm + lm_head).
//
// decode.hip's projections are split-K: a workgroup owns one 256-wide k-slab, so it reads only that slab of the operand, but every
// output collects 6-35 fp32 atomics and leaves as a RAW accumulator that the next launch finishes.  For three of the step's launches
// the other cut measures faster on MI355X (tools/probes/decode_sw_probe.cpp, profiles/r06_decode_forms.md): every output element
// has exactly ONE writer -- a workgroup owns a few weight rows for the WHOLE contraction, its six waves split K (one k-slab each,
// activation fragments in registers), the partial 16x16 MFMA tiles meet in LDS, and the finished value leaves by a plain store:
//   o projection     h += bf16(Linear(attention output))      4.7 us vs 3.7 split-K, but it leaves h FINISHED for gate/up
//   gate/up + SwiGLU act = bf16(bf16(silu(gate)) * up)         13.1 us vs 14.9 (1.7 M fp32 atomics gone; down reads a bf16 operand)
//   head slice       logits = Linear(RMSNorm(h + pending))    7.5 us vs 10.2 for the finishing kernel + split-K GEMV
// The price of this cut is operand BROADCAST: every workgroup reads the whole operand (a CU issues about one 1 KB vector-memory
// instruction per 15 ns, so the 96 KB fp32 stream costs 1.4 us per launch before the first MFMA); it loses for q/k/v (7.8 vs 5.0 us)
// and badly for the down projection (287 KB of operand per workgroup: 13.7 vs 8.4 us), which therefore stay split-K.
// RMSNorm is applied where the reference applies it: operand = bf16(w * (x * rstd)) (Qwen2RMSNorm.forward), the row statistics
// recomputed by every workgroup in a fixed order, so these launches are bit-reproducible run to run.
//   weights stream HBM -> LDS by LDS-DMA (nt), 16 rows x 256 k per tile, source-side XOR swizzle, per-wave ring, all vector memory
//   hand-issued with counted waits (vmem_asm.h); 256 workgroups split the 1.5B model's projections exactly (6 / 70 / 32 weight rows).
#include "common.h"
#include "unigen_hip.h"
#include "vmem_asm.h"
#include <stdlib.h>

namespace {

enum { PRO_NORM = 0, PRO_BF16 = 1 };
enum { EPI_RESID = 1, EPI_SWIGLU = 2, EPI_STORE = 3, EPI_ATOMIC = 4 };

struct SwArgs {
  const bf16_t* W; int ldw; int K; int R;
  int nunits, upw;                                         // units of work in all / per workgroup (meaning depends on the epilogue)
  const float* h; const float* norm_w; float eps;          // PRO_NORM: fp32 residual stream [R][K], RMSNorm weight
  const float* pend; int ld_pend; float* x_out;            // PEND: h + float(bf16(pend)) is the residual stream (a split-K producer's raw
                                                           //   accumulator); the first eight workgroups write it to x_out (!= h) if given
  const bf16_t* xb; int ldx;                               // PRO_BF16: finished bf16 operand [R][ldx]
  float* h_io;                                             // EPI_RESID: h[r][col] += float(bf16(y))
  bf16_t* act; int ld_act; int I;                          // EPI_SWIGLU
  float* out; int ld_out;                                  // EPI_STORE; EPI_ATOMIC: out[r][col] += y (fp32 atomics, one per k-block)
  float* zero0; float* zero1; float* ss_zero;              // EPI_ATOMIC: accumulators this launch clears (consumed by earlier launches)
  int n0_4, per0, n1_4, per1;                              //   float4 counts and every workgroup's share of them
  int* pos_inc; int* len_inc;                              // advanced by workgroup 0 when given (the step's last reader of pos is behind us)
};

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ float silu_bf(float g) { return bf2f(f2bf(g / (1.f + __expf(-g)))); }

// Workgroup -> weight rows.  A tile = 16 slots (the MFMA's B columns); `units` are what a workgroup owns:
//   EPI_RESID / EPI_STORE: unit = weight row, 16 per tile.
//   EPI_SWIGLU: unit = hidden unit c, 8 per tile: slots 0-7 = gate rows c, slots 8-15 = up rows I + c of the same eight units (gate and
//               up of a unit meet in one wave after the reduction).
template <int EPI>
struct RowMap {
  static constexpr int UPT = EPI == EPI_SWIGLU ? 8 : 16;                                 // units per tile
  static __device__ __forceinline__ int rel_row(int I, int s) {                          // weight row of slot s relative to the tile's first
    if constexpr (EPI == EPI_SWIGLU) return s < 8 ? s : I + (s - 8);
    else return s;
  }
  static __device__ __forceinline__ int nvalid(int u0, int u1, int t) { return min(max(u1 - (u0 + UPT * t), 0), UPT); }
  static __device__ __forceinline__ bool slot_valid(int s, int nv) {
    if constexpr (EPI == EPI_SWIGLU) return (s & 7) < nv;
    else return s < nv;
  }
};

// NW waves per workgroup, wave w owns k-slab w (K == 256 NW) and walks MAXT tiles of 16 weight rows through a RING-slot LDS ring of
// MAXI KiB slots.  The whole ring is requested before the operand prologue (staging only part of it first measured no faster).
// KBLK (split-K in k-blocks of NW slabs, grid.y = k-block): the partial tile of a workgroup joins the accumulator by ONE atomic per
// element (EPI_ATOMIC) -- the down projection, whose 287 KB operand rules the whole-K cut out: 5 atomics per output instead of 35.
template <int PRO, int EPI, int NW, int MAXT, int RING, int MAXI, bool PEND = false, bool KBLK = false>
__global__ __launch_bounds__(64 * NW) void gemv_sw_kernel(const bf16_t* W_, const void* opnd_, const void* aux_, int ldw_, int K_, int R_, int nunits_,
                                                          int upw_, int ld_opnd_, int I_, float eps_, SwArgs a) {
  // Argument order: the 14 dwords in front are everything the launch's first loads need (operand, epilogue input, weight tiles); built with
  // kernarg preload (Makefile) they arrive in SGPRs with the wave -- no scalar-load round trip ahead of the first vector load (0.18 us of a
  // ~4 us launch).  opnd = the fp32 stream (PRO_NORM) or the bf16 operand; aux = norm weight (PRO_NORM) or the stream to finish (EPI_RESID).
  // The rest of SwArgs is fetched in ONE batch (one asm statement naming every field: vmem_asm.h, argument hoisting) where its latency hides.
  struct {
    const bf16_t* W; int ldw, K, R, nunits, upw; const float* h; const float* norm_w; float eps; const bf16_t* xb; int ldx; float* h_io; int I;
  } f = {W_, ldw_, K_, R_, nunits_, upw_, (const float*)opnd_, (const float*)aux_, eps_, (const bf16_t*)opnd_, ld_opnd_, (float*)const_cast<void*>(aux_), I_};
  asm volatile("" ::"s"(W_), "s"(opnd_), "s"(aux_), "s"(ldw_), "s"(K_), "s"(R_), "s"(nunits_), "s"(upw_), "s"(ld_opnd_), "s"(I_));
  static_assert(KBLK == (EPI == EPI_ATOMIC) && (!KBLK || PRO == PRO_BF16), "k-blocks accumulate; whole-K launches store");
  static_assert(RING <= MAXT && MAXT <= RING * MAXI, "ring too small for the reduction image");
  static_assert(MAXT <= NW, "wave t finishes tile t");
  static_assert(!PEND || PRO == PRO_NORM, "a pending accumulator joins the fp32 stream only");
  __shared__ __attribute__((aligned(1024))) char tile[NW][RING][MAXI * 1024];
  __shared__ __attribute__((aligned(16))) float wn[PRO == PRO_NORM ? NW : 1][256];
  __shared__ float ssp[PRO == PRO_NORM ? NW : 1][16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, slot = lane & 15;
  auto rest_of_args = [&] {                                // batch 2
    if constexpr (EPI == EPI_ATOMIC)
      asm volatile("" ::"s"(a.out), "s"(a.ld_out), "s"(a.zero0), "s"(a.zero1), "s"(a.ss_zero), "s"(a.n0_4), "s"(a.per0), "s"(a.n1_4), "s"(a.per1),
                   "s"(gridDim.x));                        // (gridDim: an implicit argument)
    else if constexpr (EPI == EPI_SWIGLU && PEND) asm volatile("" ::"s"(a.pend), "s"(a.ld_pend), "s"(a.x_out), "s"(a.act), "s"(a.ld_act));
    else if constexpr (EPI == EPI_SWIGLU) asm volatile("" ::"s"(a.act), "s"(a.ld_act));
    else if constexpr (EPI == EPI_STORE && PEND)
      asm volatile("" ::"s"(a.pend), "s"(a.ld_pend), "s"(a.x_out), "s"(a.out), "s"(a.ld_out), "s"(a.pos_inc), "s"(a.len_inc));
    else if constexpr (EPI == EPI_STORE) asm volatile("" ::"s"(a.out), "s"(a.ld_out), "s"(a.pos_inc), "s"(a.len_inc));
  };
  // EPI_ATOMIC: the clears this launch carries (accumulators consumed by earlier launches).  They go out BEHIND the operand loads and the
  // first weight tiles (their arguments are not among the preloaded ones, and stores between those loads would delay them); a thread
  // stores at most a few float4: the counted waits below over-wait by that many tile instructions at worst (vmem_asm.h, rule 1).
  auto clears = [&] {
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    const int bid = blockIdx.y * gridDim.x + blockIdx.x, tid = threadIdx.x, nt = 64 * NW;
    if (a.zero0) { const int lo = bid * a.per0, hi = min(a.n0_4, lo + a.per0); for (int i = lo + tid; i < hi; i += nt) reinterpret_cast<float4*>(a.zero0)[i] = z; }
    if (a.zero1) { const int lo = bid * a.per1, hi = min(a.n1_4, lo + a.per1); for (int i = lo + tid; i < hi; i += nt) reinterpret_cast<float4*>(a.zero1)[i] = z; }
    if (a.ss_zero && bid == 0 && tid < 32) a.ss_zero[tid] = 0.f;
  };
  const int u0 = blockIdx.x * f.upw, u1 = min(f.nunits, u0 + f.upw);
  if (u0 >= u1) return;
  const int K = f.K;
  const int arow = min(slot, f.R - 1);                       // the MFMA's A row of this lane (rows past R repeat the last one)
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int slab = (KBLK ? (int)blockIdx.y * NW : 0) + wave_u;         // this wave's 256-wide k-slab
                                                                       // (the host admits only K that is a whole number of k-blocks)
  typedef RowMap<EPI> RM;

  // ---- what the epilogue needs from memory is requested before anything else: the finishing wave of tile t (wave t) would otherwise
  // start a dependent round trip when everything else is done
  float hold[4] = {0.f, 0.f, 0.f, 0.f};
  if constexpr (EPI == EPI_RESID) {
    if (wave < MAXT) {
      const int col = min(u0 + 16 * wave + slot, u1 - 1);
#pragma unroll
      for (int j = 0; j < 4; ++j) hold[j] = f.h_io[__umul24(min(g * 4 + j, f.R - 1), f.nunits) + col];
    }
  }

  // ---- operand loads (the critical chain), then the weight DMA
  f32x4_t xa[PRO == PRO_NORM ? 8 : 1][2], pa[PEND ? 8 : 1][2];
  f32x4_t wv;
  bf16x8_t xf[8];
  if constexpr (PRO == PRO_NORM) {
    const uint32_t xo = (uint32_t)(__umul24(arow, K) + wave * 256 + g * 8) * 4u;
#define UG_LDX(u) ld16<(u) * 128>(xa[u][0], (uint64_t)f.h, xo); ld16<(u) * 128 + 16>(xa[u][1], (uint64_t)f.h, xo);
    UG_LDX(0) UG_LDX(1) UG_LDX(2) UG_LDX(3) UG_LDX(4) UG_LDX(5) UG_LDX(6) UG_LDX(7)
#undef UG_LDX
    if constexpr (PEND) {
      rest_of_args();                                        // (behind the stream's own loads)
      const uint32_t po = (uint32_t)(__umul24(arow, a.ld_pend) + wave * 256 + g * 8) * 4u;
#define UG_LDX(u) ld16<(u) * 128>(pa[u][0], (uint64_t)a.pend, po); ld16<(u) * 128 + 16>(pa[u][1], (uint64_t)a.pend, po);
      UG_LDX(0) UG_LDX(1) UG_LDX(2) UG_LDX(3) UG_LDX(4) UG_LDX(5) UG_LDX(6) UG_LDX(7)
#undef UG_LDX
    }
    ld16<0>(wv, (uint64_t)f.norm_w, (uint32_t)(wave * 256 + lane * 4) * 4u);
  } else {
    const uint32_t xo = (uint32_t)(__umul24(arow, f.ldx) + slab * 256 + g * 8) * 2u;
#define UG_LDX(u) ld16<(u) * 64>(xf[u], (uint64_t)f.xb, xo);
    UG_LDX(0) UG_LDX(1) UG_LDX(2) UG_LDX(3) UG_LDX(4) UG_LDX(5) UG_LDX(6) UG_LDX(7)
#undef UG_LDX
  }
  // byte offset of this lane inside a tile for DMA instruction i (slots 2i, 2i + 1 x 512 bytes; the bank swizzle -- 16-byte chunk
  // index ^ slot -- is applied on the SOURCE side, LDS-DMA writes lane-linear): the same for every tile of the workgroup
  uint32_t voff[MAXI];
#pragma unroll
  for (int i = 0; i < MAXI; ++i) {
    const int sl = 2 * i + (lane >> 5);
    voff[i] = (uint32_t)(RM::rel_row(f.I, sl) * f.ldw + (((lane & 31) ^ sl) << 3)) * 2u;
  }
  // stage tile t into ring slot t % RING.  ALWAYS MAXI instructions: the waits below count instructions.  A slot without a weight row
  // reads the tile's first 16 bytes (one line for the whole wave) into LDS nobody uses.
  auto stage = [&](int t) {
    const int nv = RM::nvalid(u0, u1, t);
    const int r0 = nv > 0 ? u0 + RM::UPT * t : u0;
    const uint64_t base = (uint64_t)f.W + ((int64_t)r0 * f.ldw + slab * 256) * 2;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_addr_of(tile[wave][t % RING]));
#pragma unroll
    for (int i = 0; i < MAXI; ++i)       // nt: every weight byte is read once per step by one CU (guide, price list row nt-weights)
      dma16_nt(base, RM::slot_valid(2 * i + (lane >> 5), nv) ? voff[i] : 0u, dst + i * 1024);
  };
#pragma unroll
  for (int t = 0; t < RING; ++t) stage(t);
  if constexpr (!PEND) rest_of_args();                        // (the epilogue's arguments: their round trip hides under the operand's)
  if constexpr (EPI == EPI_ATOMIC) clears();

  wait_vm<RING * MAXI>();                                      // the operand loads are older than every DMA instruction
  if constexpr (PRO == PRO_NORM) {
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
        if (a.x_out && blockIdx.x == u && slot < f.R) {         // eight workgroups x six waves x one k-step cover the whole stream
          float* xo = a.x_out + (__umul24(slot, K) + wave * 256 + u * 32 + g * 8);
          *reinterpret_cast<f32x4_t*>(xo) = xa[u][0];
          *reinterpret_cast<f32x4_t*>(xo + 4) = xa[u][1];
        }
      }
    }
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
    lds_barrier();
    float tot = 0.f;
#pragma unroll
    for (int ww = 0; ww < NW; ++ww) tot += ssp[ww][slot];      // same order in every workgroup: one value of rstd per row everywhere
    const float rs = rsqrtf(tot / (float)K + f.eps);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float4 w0 = *reinterpret_cast<const float4*>(&wn[wave][u * 32 + g * 8]);
      const float4 w1 = *reinterpret_cast<const float4*>(&wn[wave][u * 32 + g * 8 + 4]);
      // Qwen2RMSNorm.forward: weight * (x * rsqrt(mean(x^2) + eps)), then the Linear's autocast rounds its input to bf16
      const uint32_t p0 = pack_bf2(w0.x * (xa[u][0][0] * rs), w0.y * (xa[u][0][1] * rs));
      const uint32_t p1 = pack_bf2(w0.z * (xa[u][0][2] * rs), w0.w * (xa[u][0][3] * rs));
      const uint32_t p2 = pack_bf2(w1.x * (xa[u][1][0] * rs), w1.y * (xa[u][1][1] * rs));
      const uint32_t p3 = pack_bf2(w1.z * (xa[u][1][2] * rs), w1.w * (xa[u][1][3] * rs));
      xf[u] = __builtin_bit_cast(bf16x8_t, make_uint4(p0, p1, p2, p3));
    }
  } else {
#pragma unroll
    for (int u = 0; u < 8; ++u) tie(xf[u]);
  }

  f32x4_t acc[MAXT];
#pragma unroll
  for (int t = 0; t < MAXT; ++t) acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < MAXT; ++t) {
    // DMA instructions issued behind tile t's own when its fragments are read: tiles t + 1 .. min(MAXT, t + RING) - 1
    wait_vm_n(((t + RING < MAXT ? t + RING : MAXT) - 1 - t) * MAXI);
    const char* tr = tile[wave][t % RING] + slot * 512;
    bf16x8_t wf[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) wf[u] = *reinterpret_cast<const bf16x8_t*>(tr + (((u * 4 + g) ^ slot) << 4));
    if (t + RING < MAXT) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this slot's fragments are in registers before the DMA overwrites it
      stage(t + RING);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[u], wf[u], acc[t], 0, 0, 0);
  }
  // ---- the NW partial tiles meet in LDS (each wave parks its own in its OWN ring area: nobody else reads that), fixed order
  {
    f32x4_t* red = reinterpret_cast<f32x4_t*>(&tile[wave][0][0]);
#pragma unroll
    for (int t = 0; t < MAXT; ++t) red[t * 64 + lane] = acc[t];
  }
  lds_barrier();
  if (a.pos_inc && blockIdx.x == 0 && threadIdx.x == 0) { ++*a.pos_inc; ++*a.len_inc; }
  const int t = wave;                                        // wave t finishes tile t
  if (t < MAXT && RM::nvalid(u0, u1, t) > 0) {
    f32x4_t v = reinterpret_cast<const f32x4_t*>(&tile[0][0][0])[t * 64 + lane];
#pragma unroll
    for (int ww = 1; ww < NW; ++ww) v += reinterpret_cast<const f32x4_t*>(&tile[ww][0][0])[t * 64 + lane];
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
          if (r < f.R) a.act[__umul24(r, a.ld_act) + unit] = f2bf(silu_bf(bf2f(f2bf(v[j]))) * bf2f(f2bf(up[j])));
        }
      }
    } else if constexpr (EPI == EPI_RESID) {
      const int col = u0 + 16 * t + slot;
      if (col < u1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = g * 4 + j;
          if (r < f.R) f.h_io[__umul24(r, f.nunits) + col] = hold[j] + bf2f(f2bf(v[j]));
        }
      }
    } else {
      const int col = u0 + 16 * t + slot;
      if (col < u1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = g * 4 + j;
          if (r < f.R) {
            if constexpr (EPI == EPI_ATOMIC) atomicAdd(a.out + ((int64_t)r * a.ld_out + col), v[j]);
            else a.out[(int64_t)r * a.ld_out + col] = v[j];
          }
        }
      }
    }
  }
}

int cu_count() {
  static const int n = [] {
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

// the kernel's leading scalar arguments (kernarg preload: see gemv_sw_kernel) out of a filled SwArgs
#define UG_SW_LEAD_NORM(a) (a).W, (const void*)(a).h, (const void*)(a).norm_w, (a).ldw, (a).K, (a).R, (a).nunits, (a).upw, 0, (a).I, (a).eps
#define UG_SW_LEAD_BF16(a, aux) (a).W, (const void*)(a).xb, (const void*)(aux), (a).ldw, (a).K, (a).R, (a).nunits, (a).upw, (a).ldx, 0, 0.f
#define UG_SW_COMMON(name)                                                                                                                   \
  UG_REQUIRE(R > 0 && R <= 16 && W && ldw % 8 == 0 && ug_aligned16(W) && N > 0 && (int64_t)N * ldw < (1ll << 31) && ldw < (1 << 24),        \
             name ": need 1 <= rows <= 16, 16-byte aligned weight rows, N * ldw < 2^31 (rows=%ld N=%ld ldw=%ld)", (long)R, (long)N, (long)ldw)
#define UG_SW_PEND_ARGS(name)                                                                                                         \
  UG_REQUIRE(pend == nullptr || (ld_pend >= H && ld_pend % 4 == 0 && ld_pend < (1 << 20) && ug_aligned16(pend) && x_out != h &&           \
                                 (x_out == nullptr || ug_aligned16(x_out))),                                                           \
             name ": pending accumulator must be 16-byte aligned fp32 rows and x_out a buffer other than h");                         \
  UG_REQUIRE(pend != nullptr || x_out == nullptr, name ": x_out is written only together with a pending accumulator");               \
  a.pend = pend; a.ld_pend = (int)ld_pend; a.x_out = x_out

}  // namespace

extern "C" int ug_decode_sw_supported(int64_t hidden, int64_t inter, int64_t q_dim, int head_dim) {
  return hidden == 1536 && q_dim == 1536 && inter > 0 && head_dim == 128 ? 1 : 0;
}

extern "C" int ug_decode_sw_gate_up(const float* h, const float* pend, int64_t ld_pend, float* x_out, const float* norm_w, float eps,
                                    int64_t R, int64_t H, const void* W, int64_t ldw, int64_t I, void* act, int64_t ld_act,
                                    hipStream_t st) {
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
  UG_REQUIRE(!x_out || grid >= 8, "ug_decode_sw_gate_up: fewer than eight workgroups cannot write x_out");
  if (pend) hipLaunchKernelGGL((gemv_sw_kernel<PRO_NORM, EPI_SWIGLU, 6, 5, 3, 8, true>), dim3(grid), dim3(64 * 6), 0, st, UG_SW_LEAD_NORM(a), a);
  else hipLaunchKernelGGL((gemv_sw_kernel<PRO_NORM, EPI_SWIGLU, 6, 5, 3, 8>), dim3(grid), dim3(64 * 6), 0, st, UG_SW_LEAD_NORM(a), a);
  UG_CHECK_LAUNCH("ug_decode_sw_gate_up");
  return UG_OK;
}

extern "C" int ug_decode_sw_resid(const void* x, int64_t ldx, int64_t R, const void* W, int64_t ldw, int64_t N, int64_t K, float* h,
                                  hipStream_t st) {
  UG_SW_COMMON("ug_decode_sw_resid");
  UG_REQUIRE(x && h && K == 1536 && ldx >= K && ldx % 8 == 0 && ldx < (1 << 20) && ldw >= K && ug_aligned16(x) && N < (1 << 20),
             "ug_decode_sw_resid: bad args (the contraction must be 1536 wide: six waves x one 256-wide k-slab; K=%ld)", (long)K);
  SwArgs a{};
  a.W = (const bf16_t*)W; a.ldw = (int)ldw; a.K = (int)K; a.R = (int)R;
  a.nunits = (int)N; a.xb = (const bf16_t*)x; a.ldx = (int)ldx; a.h_io = h;
  a.upw = units_per_wg(a.nunits, 8);
  const unsigned grid = (unsigned)((a.nunits + a.upw - 1) / a.upw);
  hipLaunchKernelGGL((gemv_sw_kernel<PRO_BF16, EPI_RESID, 6, 1, 1, 4>), dim3(grid), dim3(64 * 6), 0, st, UG_SW_LEAD_BF16(a, a.h_io), a);
  UG_CHECK_LAUNCH("ug_decode_sw_resid");
  return UG_OK;
}

extern "C" int ug_decode_sw_kblock(const void* x, int64_t ldx, int64_t R, const void* W, int64_t ldw, float* acc, int64_t ldacc, int64_t N,
                                   int64_t K, float* zero0, int64_t n0, float* zero1, int64_t n1, float* ss_zero, hipStream_t st) {
  UG_SW_COMMON("ug_decode_sw_kblock");
  UG_REQUIRE(x && acc && K > 0 && K % (256 * 7) == 0 && ldx >= K && ldx % 8 == 0 && ldx < (1 << 20) && ldw >= K && ug_aligned16(x) &&
                 ldacc >= N && n0 % 4 == 0 && n1 % 4 == 0 && n0 < (1ll << 31) && n1 < (1ll << 31) && ug_aligned16(zero0) && ug_aligned16(zero1),
             "ug_decode_sw_kblock: bad args (the contraction must be a whole number of 1792-wide k-blocks: seven waves x one 256-wide "
             "k-slab; K=%ld)", (long)K);
  SwArgs a{};
  a.W = (const bf16_t*)W; a.ldw = (int)ldw; a.K = (int)K; a.R = (int)R;
  a.nunits = (int)N; a.xb = (const bf16_t*)x; a.ldx = (int)ldx; a.out = acc; a.ld_out = (int)ldacc;
  a.upw = 32;                                                  // two 16-row tiles per workgroup
  const dim3 grid((unsigned)((a.nunits + a.upw - 1) / a.upw), (unsigned)(K / (256 * 7)));
  const unsigned nblocks = grid.x * grid.y;
  a.zero0 = zero0; a.zero1 = zero1; a.ss_zero = ss_zero;
  a.n0_4 = (int)(n0 >> 2); a.n1_4 = (int)(n1 >> 2);
  a.per0 = (int)((a.n0_4 + nblocks - 1) / nblocks); a.per1 = (int)((a.n1_4 + nblocks - 1) / nblocks);
  hipLaunchKernelGGL((gemv_sw_kernel<PRO_BF16, EPI_ATOMIC, 7, 2, 2, 8, false, true>), grid, dim3(64 * 7), 0, st, UG_SW_LEAD_BF16(a, nullptr), a);
  UG_CHECK_LAUNCH("ug_decode_sw_kblock");
  return UG_OK;
}

extern "C" int ug_decode_sw_head(const float* h, const float* pend, int64_t ld_pend, float* x_out, const float* norm_w, float eps,
                                 int64_t R, int64_t H, const void* W, int64_t ldw, int64_t N, float* logits, int64_t ld_logits,
                                 int* pos_inc, int* len_inc, hipStream_t st) {
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
  UG_REQUIRE(!x_out || grid >= 8, "ug_decode_sw_head: fewer than eight workgroups cannot write x_out");
  if (pend) hipLaunchKernelGGL((gemv_sw_kernel<PRO_NORM, EPI_STORE, 6, 2, 2, 8, true>), dim3(grid), dim3(64 * 6), 0, st, UG_SW_LEAD_NORM(a), a);
  else hipLaunchKernelGGL((gemv_sw_kernel<PRO_NORM, EPI_STORE, 6, 2, 2, 8>), dim3(grid), dim3(64 * 6), 0, st, UG_SW_LEAD_NORM(a), a);
  UG_CHECK_LAUNCH("ug_decode_sw_head");
  return UG_OK;
}
