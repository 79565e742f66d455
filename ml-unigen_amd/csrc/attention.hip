// Masked flash attention (fwd + bwd) for the Qwen2.5-1.5B backbone on gfx950:
// GQA 12 query heads : 2 kv heads, head_dim 128, L <= a few thousand, bf16 operands, fp32 softmax.
// Replaces torch SDPA inside transformers' Qwen2Attention (modeling_qwen2.py:176-234) for the masks
// the reference's builders produce (training/prompting_utils.py:975-1074): the caller's dense additive
// [B,1,L,L] mask (0 = attend, huge negative = blocked) is compressed once per step into a bitmask
// (one 64-bit word per query row per 64-key tile) plus per-tile "anything visible" flags, so fully
// blocked tiles are skipped and partially blocked ones cost one bit test per score.
//
// Kernel shape (all three kernels): 4 waves x 16 rows = 64-row tiles, 64-wide tiles on the other
// axis, mfma_f32_16x16x32_bf16.  Scores are produced TRANSPOSED (rows = the streamed axis) so that
// each lane owns one query (resp. key) column: row statistics are lane-local plus two shuffles, and
// the probabilities feed the second MFMA as a B operand straight from registers.  The second
// contraction needs the streamed operand k-major: its fragments are read from the SAME row-major LDS
// tile with the hardware transposing read (ds_read_b64_tr_b16), so every operand is staged once, row-major,
// with 16-byte loads, and no transposed copy exists in HBM or LDS.
#include "common.h"
#include "unigen_hip.h"
#include "vmem_asm.h"
#include <stdlib.h>

namespace {

constexpr int HD = 128;      // head_dim
// LDS row stride (elements) of a row-major [64][128] tile: 288 B = 8 banks past a multiple of the 64-bank row, so the
// 16 rows x 16 B of a ds_read_b128 fragment fetch and the 8 rows x 32 B a ds_read_b64_tr_b16 half-wave touches both
// fall on distinct banks
constexpr int RM_LD = 144;
constexpr int RM_BYTES = 64 * RM_LD * 2;    // 18432
typedef __attribute__((ext_vector_type(4))) short s16x4_t;

// ------------------------------------------------------------------ mask compression
template <typename T> __device__ __forceinline__ bool mask_attend(T v, int* err);
template <> __device__ __forceinline__ bool mask_attend<float>(float v, int* err) {
  if (v == 0.f) return true;
  if (!(v <= -1e9f) && err) atomicOr(err, 2);   // neither 0 nor "blocked": not a mask we implement
  return false;
}
template <> __device__ __forceinline__ bool mask_attend<bf16_t>(bf16_t v, int* err) { return mask_attend<float>(bf2f(v), err); }
template <> __device__ __forceinline__ bool mask_attend<int64_t>(int64_t v, int* err) {
  if (v == 0) return true;
  if (v > -1000000000LL && err) atomicOr(err, 2);
  return false;
}
template <> __device__ __forceinline__ bool mask_attend<uint8_t>(uint8_t v, int*) { return v != 0; }   // bool: True = attend

// one wave per (batch, row, word); mask element (b, row, col) at mask[b*sb + row*sr + col]
template <typename T>
__global__ __launch_bounds__(256) void mask_compress_kernel(const T* __restrict__ mask, int64_t sb, int64_t sr,
                                                            uint64_t* __restrict__ bits, int B, int L, int nW,
                                                            int* __restrict__ err) {
  const int64_t wid = blockIdx.x * 4LL + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int64_t total = (int64_t)B * L * nW;
  if (wid >= total) return;
  const int w = (int)(wid % nW);
  const int row = (int)((wid / nW) % L);
  const int b = (int)(wid / ((int64_t)nW * L));
  const int col = w * 64 + lane;
  bool a = false;
  if (col < L) a = mask_attend<T>(mask[b * sb + row * sr + col], err);
  const uint64_t word = __ballot(a);
  if (lane == 0) bits[wid] = word;
}

// causal (optionally with a [B,L] key-validity vector): what HF applies when no 4-D mask is given
__global__ __launch_bounds__(256) void mask_causal_kernel(uint64_t* __restrict__ bits, const uint8_t* __restrict__ key_valid,
                                                          int B, int L, int nW) {
  const int64_t wid = blockIdx.x * 4LL + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int64_t total = (int64_t)B * L * nW;
  if (wid >= total) return;
  const int w = (int)(wid % nW);
  const int row = (int)((wid / nW) % L);
  const int b = (int)(wid / ((int64_t)nW * L));
  const int col = w * 64 + lane;
  bool a = (col < L) && (col <= row);
  if (a && key_valid) a = key_valid[(int64_t)b * L + col] != 0;
  const uint64_t word = __ballot(a);
  if (lane == 0) bits[wid] = word;
}

// ---- masks straight from token ids (no dense [B,1,L,L] tensor on the way) ----------------------------------------
// Reference builders (training/prompting_utils.py:975-1036; verified semantics in SURVEY.md section 8a):
//   mode 0  create_attention_mask_predict_next(rm_pad_in_image=True)  (t2i rows)
//   mode 1  create_attention_mask_predict_next()                      (lm rows)
//   mode 2  create_attention_mask_for_mmu                             (eoi position of the FIRST match in the batch)
// meta[b] = {last_pad, soi_pos, first_eoi_of_batch(+1, 0 = none), unused}; flags[b][t]: bit0 = pad, bit1 = inside <soi>..<eoi>
__global__ __launch_bounds__(64) void mask_ids_meta_kernel(const int64_t* __restrict__ ids, int L, int64_t pad_id, int64_t soi_id,
                                                           int64_t eoi_id, int* __restrict__ meta, uint8_t* __restrict__ flags) {
  // one wave per row, 64 tokens per trip: the delimiter / padding tests become ballots, the inclusive cumulative counts a
  // popcount of the ballot below each lane plus the running totals of the earlier trips (round 3 walked the row on one lane: 96 us
  // for 16 x 771 ids)
  const int b = blockIdx.x, lane = threadIdx.x;
  const int64_t* row = ids + (int64_t)b * L;
  int starts = 0, ends = 0, last_pad = -1, soi_pos = -1, eoi_first = -1;
  const uint64_t below = lane == 63 ? ~0ull : ((1ull << (lane + 1)) - 1ull);        // lanes 0 .. lane (inclusive)
  for (int t0 = 0; t0 < L; t0 += 64) {
    const int t = t0 + lane;
    const bool in = t < L;
    const int64_t v = in ? row[t] : 0;
    const bool st = in && v == soi_id, en = in && v == eoi_id, pd = in && v == pad_id;
    const uint64_t ms = __ballot(st), me = __ballot(en), mp = __ballot(pd);
    const int cs = starts + __popcll(ms & below), ce = ends + __popcll(me & below);
    const bool in_img = (cs > ce) || st || en;          // inclusive cumsum(start) > cumsum(end), or a delimiter itself
    if (in) flags[(int64_t)b * L + t] = (uint8_t)((pd ? 1 : 0) | (in_img ? 2 : 0));
    starts += __popcll(ms);
    ends += __popcll(me);
    if (mp) last_pad = t0 + 63 - __clzll(mp);
    if (ms && soi_pos < 0) soi_pos = t0 + __ffsll((long long)ms) - 1;
    if (me && eoi_first < 0) eoi_first = t0 + __ffsll((long long)me) - 1;
  }
  if (lane == 0) {
    meta[b * 4 + 0] = last_pad;
    meta[b * 4 + 1] = soi_pos < 0 ? 0 : soi_pos;        // argmax of an all-zero row is 0
    meta[b * 4 + 2] = eoi_first;
  }
}

__global__ __launch_bounds__(256) void mask_ids_kernel(uint64_t* __restrict__ bits, const int* __restrict__ meta,
                                                       const uint8_t* __restrict__ flags, int B, int L, int nW, int mode) {
  const int64_t wid = blockIdx.x * 4LL + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int64_t total = (int64_t)B * L * nW;
  if (wid >= total) return;
  const int w = (int)(wid % nW);
  const int row = (int)((wid / nW) % L);
  const int b = (int)(wid / ((int64_t)nW * L));
  const int col = w * 64 + lane;
  bool a = false;
  if (col < L) {
    if (mode == 2) {
      int eoi = -1;
      for (int bb = 0; bb < B && eoi < 0; ++bb) eoi = meta[bb * 4 + 2];       // first match in row-major order
      a = (col <= row) || (col <= eoi);
    } else {
      const uint8_t fr = flags[(int64_t)b * L + row], fc = flags[(int64_t)b * L + col];
      const bool img_row = fr & 2;
      a = img_row || (col <= row);
      if (mode == 0) {
        const int last_pad = meta[b * 4 + 0], soi_pos = meta[b * 4 + 1];
        if (!img_row && row > last_pad && col <= last_pad) a = false;        // text after the padding never looks at it
        if (img_row && row >= soi_pos && (fc & 1)) a = false;                // image rows never look at pad columns
      }
    }
  }
  const uint64_t word = __ballot(a);
  if (lane == 0) bits[wid] = word;
}

// tileany[b, qt, w] = OR over the 64 rows of q-tile qt of bits[b, row, w] != 0 ; one wave each
__global__ __launch_bounds__(256) void mask_tiles_kernel(const uint64_t* __restrict__ bits, uint8_t* __restrict__ tileany,
                                                         int B, int L, int nW) {
  const int64_t wid = blockIdx.x * 4LL + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int nQ = nW;
  const int64_t total = (int64_t)B * nQ * nW;
  if (wid >= total) return;
  const int w = (int)(wid % nW);
  const int qt = (int)((wid / nW) % nQ);
  const int b = (int)(wid / ((int64_t)nW * nQ));
  const int row = qt * 64 + lane;
  uint64_t word = 0;
  if (row < L) word = bits[((int64_t)b * L + row) * nW + w];
  const uint64_t any = __ballot(word != 0);
  if (lane == 0) tileany[wid] = any ? 1 : 0;
}

// ------------------------------------------------------------------ tile staging helpers
// row-major tile: 64 rows x 128 cols from global rows (row_first + r, clamped to L-1)
template <int NT = 256>
__device__ __forceinline__ void stage_rm(bf16_t* dst, const bf16_t* src_seq, int64_t ld, int row_first, int L, int tid) {
#pragma unroll
  for (int i = 0; i < 1024 / NT; ++i) {
    const int c = tid + i * NT;
    const int r = c >> 4, ch = c & 15;
    const int gr = min(row_first + r, L - 1);
    const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(src_seq + (int64_t)gr * ld + ch * 8);
    *reinterpret_cast<bf16x8_t*>(dst + r * RM_LD + ch * 8) = v;
  }
}
// MFMA A-fragment (16 rows x 32 k) from a row-major tile: rows rb*16.., k-step ks
__device__ __forceinline__ bf16x8_t frag_rm(const bf16_t* t, int rb, int ks, int lane) {
  return *reinterpret_cast<const bf16x8_t*>(t + (rb * 16 + (lane & 15)) * RM_LD + ks * 32 + (lane >> 4) * 8);
}
// MFMA A-fragment (16 rows x 32 k) of the TRANSPOSE of a row-major tile t[64 streamed rows][128 d]: fragment rows are
// d = rb*16 .. rb*16+15, the contraction runs over the tile's rows with the k index mapped as
// slot s of lane-group g  <->  tile row  jp*32 + (s>>2)*16 + g*4 + (s&3)   (matches the C-layout of two adjacent 16-wide
// score blocks, so probabilities feed the B operand without a shuffle).  ds_read_b64_tr_b16: within a 16-lane group the
// lanes 4j..4j+3 address the four 8-byte pieces of tile row j (16 d values), and lane i receives column i of that 4 x 16
// block -- the four streamed rows of d = rb*16 + i.
__device__ __forceinline__ bf16x8_t frag_trr(const bf16_t* t, int rb, int jp, int lane) {
  const int i16 = lane & 15, g = lane >> 4;
  const bf16_t* p0 = t + (jp * 32 + g * 4 + (i16 >> 2)) * RM_LD + rb * 16 + (i16 & 3) * 4;
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p0);
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p0 + 16 * RM_LD));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ bf16x8_t pack_p(const f32x4_t& a, const f32x4_t& b) {
  bf16x8_t r;
  r[0] = (short)f2bf(a[0]); r[1] = (short)f2bf(a[1]); r[2] = (short)f2bf(a[2]); r[3] = (short)f2bf(a[3]);
  r[4] = (short)f2bf(b[0]); r[5] = (short)f2bf(b[1]); r[6] = (short)f2bf(b[2]); r[7] = (short)f2bf(b[3]);
  return r;
}
// integer-sequence RNE packing (what pack_p was before the hardware converter).  The dQ kernel keeps it: measured in one
// session, 216.6 us with this sequence vs 255 us with v_cvt_pk_bf16_f32 (the forward and dK/dV kernels go the other way).
__device__ __forceinline__ bf16_t f2bf_sw(float f) {
  // branch-free: the NaN test of round 2 compiled to an exec-mask branch per element (60 scalar instructions per key tile).
  // A NaN stays a NaN whatever its payload (a select, v_cndmask: adding the rounding constant to a NaN whose low payload bits
  // are set would wrap it to +-0 or a denormal and hide a divergence in dQ)
  const uint32_t u = __float_as_uint(f);
  const uint32_t r = u + 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(((u & 0x7fffffffu) > 0x7f800000u ? (u | 0x00400000u) : r) >> 16);
}
__device__ __forceinline__ bf16x8_t pack_p_sw(const f32x4_t& a, const f32x4_t& b) {
  bf16x8_t r;
  r[0] = (short)f2bf_sw(a[0]); r[1] = (short)f2bf_sw(a[1]); r[2] = (short)f2bf_sw(a[2]); r[3] = (short)f2bf_sw(a[3]);
  r[4] = (short)f2bf_sw(b[0]); r[5] = (short)f2bf_sw(b[1]); r[6] = (short)f2bf_sw(b[2]); r[7] = (short)f2bf_sw(b[3]);
  return r;
}
// reduce across the 4 lane groups (lanes l, l^16, l^32, l^48 share the same column)
__device__ __forceinline__ float group_max(float v) { v = fmaxf(v, __shfl_xor(v, 16, 64)); return fmaxf(v, __shfl_xor(v, 32, 64)); }
__device__ __forceinline__ float group_sum(float v) { v += __shfl_xor(v, 16, 64); return v + __shfl_xor(v, 32, 64); }

struct AttnArgs {
  const bf16_t* q; const bf16_t* k; const bf16_t* v;   // row (b*L+t), head h at +h*128 ; row stride ldq
  bf16_t* o; const bf16_t* dout;                       // [tokens, ldo]
  bf16_t* dq; bf16_t* dk; bf16_t* dv;                  // row stride ldg
  float* dkv_ws;                                       // [tokens][2*HKV*128] fp32, split-head dK/dV accumulation (or null)
  float* lse; const float* delta;                      // [B][H][L]
  const uint64_t* bits; const uint8_t* tileany;
  int64_t ldq, ldo, ldg;
  int B, L, Lp, nW, H, HKV;
  float scale;
  // backward only: RoPE transposed on dq / dk before they are stored, and the q | k | v bias gradient (column sums of the stored
  // bf16 values) added into dbias[(H + 2 HKV) * 128] -- what the rope and colsum passes over dqkv did (or null: plain stores)
  const float* rope_cos; const float* rope_sin; float* dbias;
  int order;                // workgroup order on an XCD: 0 pair by pair, 1 tile rank first over all pairs (wg_coord)
  int dkv_heads;            // query heads per workgroup of the split-head dK / dV kernel (1 or 2)
  int ablate;               // probe builds of the timing tools only (UNIGEN_ATTN_ABLATE): 1 = no K / V staging after the first tile
};

// RoPE backward of one rotary pair, the arithmetic of rope_kernel<true> (elementwise.hip): the gradient is rounded to bf16
// first (it arrives as a bf16 tensor in the reference), products and sums round separately
__device__ __forceinline__ void rope_bwd_pair(float& g1, float& g2, float c, float s) {
#pragma clang fp contract(off)
  const float x1 = bf2f(f2bf(g1)), x2 = bf2f(f2bf(g2));
  const float a1 = x1 * c, a2 = x2 * c;
  const float b1 = x2 * s, b2 = x1 * s;
  g1 = a1 + b1; g2 = a2 - b2;
}
// sum of x[v] over the 32 lanes n = lane & 31 (same lane half) by recursive halving: 62 exchanges instead of 320; lane n ends up
// with the sums of v = 2 n and 2 n + 1 in x[0], x[1]
template <int N>
__device__ __forceinline__ void halve_sum(float (&x)[64], int lane, int m) {
  const bool up = (lane & m) != 0;
#pragma unroll
  for (int i = 0; i < N / 2; ++i) {
    const float lo = x[i], hi = x[i + N / 2];
    x[i] = (up ? hi : lo) + __shfl_xor(up ? lo : hi, m, 64);
  }
}
__device__ __forceinline__ void lanes32_colsum(float (&x)[64], int lane) {
  halve_sum<64>(x, lane, 16); halve_sum<32>(x, lane, 8); halve_sum<16>(x, lane, 4); halve_sum<8>(x, lane, 2); halve_sum<4>(x, lane, 1);
}

// Workgroup -> (tile, head, batch).  Consecutive workgroup ids of a launch go round-robin over the 8 XCDs, each with a private
// 4 MB L2; with a (tile, head, batch) grid every XCD meets every (batch, kv head) and re-fetches ALL K / V rows (12.6 MB at
// 16 x 771 tokens) through its own L2 -- 387 MB of L2 fills per forward launch for 12.6 MB of distinct data.  Here every
// workgroup that reads the K / V rows of one (batch, kv head) pair -- its G query heads x nT tiles -- runs on ONE XCD (4 pairs
// of 395 KB per XCD at the training shape).  1-D grid of 8 * ceil(pairs / 8) * nT * G workgroups; surplus ones exit.
struct WgCoord { int tile, h, b; bool ok; };
// heavy_last: the tiles' cost grows with their index (query tiles under a causal mask) -> hand them out in descending order,
// so that the launch ends with its cheapest workgroups
__device__ __forceinline__ WgCoord wg_coord(int wid, int nT, int H, int HKV, int B, bool heavy_last = false, bool global_order = false) {
  // Order on an XCD, longest first.  global_order: tile rank first over ALL the XCD's pairs and heads, then pair, then head.
  // Otherwise pair by pair and, inside a pair, tile rank first over its G heads.  (Round 3 walked head by head inside a pair: a
  // head's light tiles were handed out before the next head's heavy ones.  Measured at 16 x 771 tokens, causal mask, us per layer:
  // forward 77 -> 69 with either new order, 116 vs 128 on the full mask for global vs pair-major; fused backward 271 -> 255
  // pair-major, 272 global.)
  const int G = H / HKV, P = B * HKV;
  const int xcd = wid & 7, idx = wid >> 3;
  WgCoord c;
  int pair, rank, head;
  if (global_order) {
    const int ppx = (P + 7) >> 3;
    rank = idx / (ppx * G);
    const int rem = idx % (ppx * G);
    pair = (rem / G) * 8 + xcd;
    head = rem % G;
  } else {
    const int per_pair = nT * G, rem = idx % per_pair;
    pair = (idx / per_pair) * 8 + xcd;
    rank = rem / G;
    head = rem % G;
  }
  c.ok = pair < P;
  c.b = pair / HKV;
  c.h = (pair % HKV) * G + head;
  c.tile = heavy_last ? nT - 1 - rank : rank;
  return c;
}
static inline unsigned wg_grid(int64_t nT, int H, int HKV, int64_t B) {
  const int64_t P = B * HKV;
  return (unsigned)(8 * ((P + 7) / 8) * nT * (H / HKV));
}

// ================================================================== forward
// grid (ceil(L / (16 NW)), H, B).  NW = 4: one 64-row query tile per workgroup; NW = 8: two (128 rows) sharing every staged
// K / V tile -- half the staging traffic and barriers per query row, same LDS, same registers per wave.  A wave whose own
// 64-row tile sees nothing of a key tile skips the arithmetic but keeps the barriers.
template <int NW>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 4) void attn_fwd_kernel(AttnArgs p) {
  __shared__ __attribute__((aligned(16))) bf16_t Ks[64 * RM_LD];
  __shared__ __attribute__((aligned(16))) bf16_t Vs[64 * RM_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
  const WgCoord wc = wg_coord((int)blockIdx.x, (p.L + 16 * NW - 1) / (16 * NW), p.H, p.HKV, p.B);
  if (!wc.ok) return;
  const int qt = wc.tile, h = wc.h, b = wc.b;
  const int hk = h / (p.H / p.HKV);
  const int qrow = qt * (16 * NW) + wave * 16 + (lane & 15);   // this lane's query (column of S^T)
  const int qrow_c = min(qrow, p.L - 1);
  const bf16_t* qseq = p.q + (int64_t)b * p.L * p.ldq + h * HD;
  const bf16_t* kseq = p.k + (int64_t)b * p.L * p.ldq + hk * HD;
  const bf16_t* vseq = p.v + (int64_t)b * p.L * p.ldq + hk * HD;

  bf16x8_t qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
    qf[ks] = *reinterpret_cast<const bf16x8_t*>(qseq + (int64_t)qrow_c * p.ldq + ks * 32 + g * 8);

  f32x4_t ot[8];
#pragma unroll
  for (int d = 0; d < 8; ++d) ot[d] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float m_i = -INFINITY, l_i = 0.f;
  const uint64_t* wrow = p.bits + ((int64_t)b * p.L + qrow_c) * p.nW;
  const int q64 = qt * (NW / 4) + (wave >> 2);                 // this wave's 64-row tile (its own row of tile flags)
  const uint8_t* tany = p.tileany + ((int64_t)b * p.nW + min(q64, p.nW - 1)) * p.nW;
  const uint8_t* tany2 = p.tileany + ((int64_t)b * p.nW + min(q64 ^ (NW == 8 ? 1 : 0), p.nW - 1)) * p.nW;   // the other half's
  const bool have = q64 < p.nW, have2 = (q64 ^ (NW == 8 ? 1 : 0)) < p.nW;

  for (int t = 0; t < p.nW; ++t) {
    const bool mine = have && tany[t];
    if (!mine && !(NW == 8 && have2 && tany2[t])) continue;     // uniform per block: both halves evaluate the same pair
    __syncthreads();
    stage_rm<64 * NW>(Ks, kseq, p.ldq, t * 64, p.L, tid);
    stage_rm<64 * NW>(Vs, vseq, p.ldq, t * 64, p.L, tid);   // rows past L repeat the last key: their probabilities are exact zeros
    __syncthreads();
    if (!mine) continue;                                        // wave-uniform: this 64-row tile sees none of these keys

    f32x4_t st[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      st[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        st[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rm(Ks, j, ks, lane), qf[ks], st[j], 0, 0, 0);
    }
    const uint64_t word = wrow[t];
    float mloc = -INFINITY;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool on = (word >> (j * 16 + g * 4 + r)) & 1ull;
        const float s = on ? st[j][r] * p.scale : -INFINITY;
        st[j][r] = s;
        mloc = fmaxf(mloc, s);
      }
    mloc = group_max(mloc);
    const float m_new = fmaxf(m_i, mloc);
    const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = __expf(m_i - m_use);
    float rs = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float e = __expf(st[j][r] - m_use); st[j][r] = e; rs += e; }
    rs = group_sum(rs);
    l_i = l_i * alpha + rs;
    m_i = m_new;
#pragma unroll
    for (int d = 0; d < 8; ++d) { ot[d][0] *= alpha; ot[d][1] *= alpha; ot[d][2] *= alpha; ot[d][3] *= alpha; }
    const bf16x8_t pf0 = pack_p(st[0], st[1]), pf1 = pack_p(st[2], st[3]);
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      ot[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_trr(Vs, d, 0, lane), pf0, ot[d], 0, 0, 0);
      ot[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_trr(Vs, d, 1, lane), pf1, ot[d], 0, 0, 0);
    }
  }
  if (qrow < p.L) {
    const float inv = (l_i > 0.f) ? 1.f / l_i : 0.f;
    bf16_t* orow = p.o + ((int64_t)b * p.L + qrow) * p.ldo + h * HD;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      uint2 w; w.x = pack_bf2(ot[d][0] * inv, ot[d][1] * inv); w.y = pack_bf2(ot[d][2] * inv, ot[d][3] * inv);
      *reinterpret_cast<uint2*>(orow + d * 16 + g * 4) = w;
    }
    if (g == 0) p.lse[((int64_t)b * p.H + h) * p.L + qrow] = (l_i > 0.f) ? m_i + __logf(l_i) : INFINITY;
  }
}

// ================================================================== forward, 32 query rows per wave (round 3)
// The same S^T formulation on v_mfma_f32_32x32x16_bf16: a wave owns 32 query rows (lane = query column n = lane & 31, key half
// hh = lane >> 5), a workgroup of NW waves 32 * NW rows.  Against the 16-row kernel above, per query row: half the LDS fragment
// reads (every K / V fragment feeds a 32-row MFMA), one cross-lane exchange per row statistic instead of two, the softmax in the
// exp2 domain (one fma + one exp per score), and the K / V tiles arrive by LDS-DMA into a two-slot ring one visible tile ahead
// with ONE workgroup barrier per tile (the 16-row kernel stages through registers between two barriers).
//   LDS tile image [64 rows][16 chunks of 16 B], chunk_l = chunk ^ swz16(row): the 16 rows of a ds_read_b128 lane group and the
//   4 keys x 4 chunks of a transposing half-wave both fall on 16 distinct 16-byte bank groups.
//   k-slot mapping of the P.V contraction (k-step j of 32-key block kb): slot e of lane half hh <-> key 16 j + 8 (e / 4) + 4 hh
//   + e % 4 -- exactly the rows the S^T accumulator holds in registers 8 j .. 8 j + 7, so probabilities feed the B operand
//   without a shuffle and the V^T fragment is two transposing reads of 4 consecutive keys each.
typedef __attribute__((ext_vector_type(8))) __bf16 hwbf16x8_t;
typedef const __attribute__((address_space(1))) void* a_gptr_t;
typedef __attribute__((address_space(3))) void* a_lptr_t;
__device__ __forceinline__ int swz16(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
constexpr int T32_BYTES = 64 * 256;
// Query tiles of the 32-row kernels are aligned to the END of the sequence: tile qt holds rows off + qt * ROWS ... with
// off = L - nT * ROWS <= 0, so the ragged tile is the FIRST one.  Every BASELINE length is 128 k + 3 or 128 k + 67 (max_seq + 256 + 3):
// aligned to the start, the last tile held 3 image rows that see every key tile and cost a full workgroup (16 x 771: forward
// 78.7 us against 71.6 at L = 768); at the front the same 3 rows are causal text rows that see one key tile.
// Tile flags (`tileany`, per 64 x 64 block) of a row range that is no longer 64-aligned: the OR over the blocks it touches.
__device__ __forceinline__ bool rows_see_tile(const uint8_t* tileany, int b, int nW, int lo, int hi, int L, int kt) {
  lo = max(lo, 0); hi = min(hi, L - 1);
  bool v = false;
  for (int q64 = lo >> 6; lo <= hi && q64 <= (hi >> 6); ++q64) v = v || tileany[((int64_t)b * nW + q64) * nW + kt];
  return v;
}

template <int NW>
__global__ __launch_bounds__(64 * NW, 2) void attn_fwd32_kernel(AttnArgs p) {
  __shared__ __attribute__((aligned(16))) char ring[2 * 2 * T32_BYTES];       // [slot][K | V]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 31, hh = lane >> 5, i16 = lane & 15, grp = (lane >> 4) & 1;
  constexpr int ROWS = 32 * NW;
  const WgCoord wc = wg_coord((int)blockIdx.x, (p.L + ROWS - 1) / ROWS, p.H, p.HKV, p.B, true, p.order != 0);
  if (!wc.ok) return;
  const int qt = wc.tile, h = wc.h, b = wc.b;
  const int hk = h / (p.H / p.HKV);
  const int row0 = p.L - ((p.L + ROWS - 1) / ROWS) * ROWS + qt * ROWS;       // end-aligned tiles: < 0 in the first one
  const int qrow = row0 + wave * 32 + n;
  const bool qlive = qrow >= 0;                                              // (always < L)
  const int qrow_c = max(qrow, 0);
  const bf16_t* qseq = p.q + (int64_t)b * p.L * p.ldq + h * HD;
  const bf16_t* kseq = p.k + (int64_t)b * p.L * p.ldq + hk * HD;
  const bf16_t* vseq = p.v + (int64_t)b * p.L * p.ldq + hk * HD;

  bf16x8_t qf[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks)
    qf[ks] = *reinterpret_cast<const bf16x8_t*>(qseq + (int64_t)qrow_c * p.ldq + ks * 16 + hh * 8);
  f32x16_t ot[4];
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[d][r] = 0.f;
  float m_i = -INFINITY, l_i = 0.f;                    // running maximum of the RAW scores, running sum
  const float c1 = p.scale * 1.4426950408889634f;      // exp(scale * s) = exp2(c1 * s)
  const uint64_t* wrow = p.bits + ((int64_t)b * p.L + qrow_c) * p.nW;
  // 64-row tile flags (nW <= 64 tiles), fetched ONCE: bit t of `vis` = some row of the workgroup sees key tile t (uniform),
  // bit t of `minem` = this wave's 32 rows do.  (A flag byte fetched per tile right before its use was a dependent global
  // load on every iteration's critical path.)
  uint64_t vis, minem;
  {
    bool v = false, m = false;
    if (lane < p.nW) {
      v = rows_see_tile(p.tileany, b, p.nW, row0, row0 + ROWS - 1, p.L, lane);
      m = rows_see_tile(p.tileany, b, p.nW, row0 + wave * 32, row0 + wave * 32 + 31, p.L, lane);
    }
    vis = __ballot(v);
    minem = __ballot(m);
  }
  auto next_visible = [&](int t) {
    const uint64_t rest = t < 64 ? vis >> t : 0ull;
    return rest ? t + __builtin_ctzll(rest) : p.nW;
  };
  // LDS-DMA of one K / V tile pair: 16 one-KiB instructions per operand, 16 / NW per wave
  // byte offset of this lane's 16 bytes inside a 64-key tile for each of its DMA instructions: computed ONCE -- recomputed per tile
  // (clamp, 64-bit row x stride product, swizzle) it was 43 VALU + ~40 SALU instructions = a sixth of the loop's issue slots
  uint32_t toff[16 / NW];
#pragma unroll
  for (int i = 0; i < 16 / NW; ++i) {
    const int row = (i * NW + wave) * 4 + (lane >> 4);
    toff[i] = (uint32_t)(row * p.ldq + (((lane & 15) ^ swz16(row)) << 3)) * 2u;
  }
  auto stage = [&](int t, int slot) {
    char* base = ring + slot * 2 * T32_BYTES;
    const int64_t tb = (int64_t)t * 64 * p.ldq * 2;
    const uint64_t kb = (uint64_t)kseq + tb, vb = (uint64_t)vseq + tb;
    // hand-issued (vmem_asm.h): behind the builtin hipcc put s_waitcnt vmcnt(0) in front of the tile reads below -- the tile
    // requested a moment ago had to LAND before the first MFMA of the current one, i.e. the ring prefetched nothing (round 6)
    if (t * 64 + 64 <= p.L) {
#pragma unroll
      for (int i = 0; i < 16 / NW; ++i) {
        const int inst = i * NW + wave;
        dma16(kb, toff[i], __builtin_amdgcn_readfirstlane(lds_addr_of(base + inst * 1024)));
        dma16(vb, toff[i], __builtin_amdgcn_readfirstlane(lds_addr_of(base + T32_BYTES + inst * 1024)));
      }
    } else {                                             // the sequence's last, partial tile: rows past L repeat row L - 1
#pragma unroll
      for (int i = 0; i < 16 / NW; ++i) {
        const int inst = i * NW + wave;
        const int row = inst * 4 + (lane >> 4);
        const uint32_t off = (uint32_t)((min(t * 64 + row, p.L - 1) - t * 64) * p.ldq + (((lane & 15) ^ swz16(row)) << 3)) * 2u;
        dma16(kb, off, __builtin_amdgcn_readfirstlane(lds_addr_of(base + inst * 1024)));
        dma16(vb, off, __builtin_amdgcn_readfirstlane(lds_addr_of(base + T32_BYTES + inst * 1024)));
      }
    }
  };
  // per-lane fragment offsets inside a tile
  int koff[8], voff[4][2];
  {
    const int sw = swz16(n);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) koff[ks] = n * 256 + (((ks * 2 + hh) ^ sw) << 4);
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int key = 4 * hh + (i16 >> 2) + 8 * e;                    // + 16 j + 32 kb: multiples of 16 leave swz16 unchanged
        const int chunk = d * 4 + grp * 2 + ((i16 & 3) >> 1);
        voff[d][e] = key * 256 + ((chunk ^ swz16(key)) << 4) + (i16 & 1) * 8;
      }
  }

  int t = next_visible(0);
  if (t < p.nW) stage(t, 0);
  // The mask word of a tile is requested one tile ahead, like its K / V rows.  It stays a compiler-visible load (a hand-issued one
  // would be a loop-carried register in flight, which the compiler may copy before it lands); what matters is WHERE its first use
  // sits: the compiler's wait for it knows nothing of the DMA around it, so the use (tie) comes right behind the loop-top drain,
  // where nothing is in flight anyway -- anywhere later it would drain the tile just requested.
  uint64_t wcur = t < p.nW ? wrow[t] : 0ull;
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) tie(qf[ks]);            // (the query fragments' own wait falls here, ahead of the loop)
  for (int it = 0; t < p.nW; ++it) {
    const int tn = next_visible(t + 1);
    wait_vm<0>();                                        // this wave's share of tile t has landed (and its mask word)
    tie(wcur);
    asm volatile("s_barrier" ::: "memory");               // ... everyone's has; everyone is past its reads of the other slot
    if (tn < p.nW && !p.ablate) stage(tn, (it + 1) & 1);
    const uint64_t wnext = tn < p.nW ? wrow[tn] : 0ull;
    const bool mine = (minem >> t) & 1ull;
    if (mine) {
      const char* Ks = ring + (it & 1) * 2 * T32_BYTES;
      const char* Vs = Ks + T32_BYTES;
      f32x16_t st[2];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) st[kb][r] = 0.f;
      // the two key blocks alternate: consecutive MFMAs never wait for each other's result (a 32x32x16 issues every 32 clocks
      // and delivers after 64)
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(Ks + kb * 8192 + koff[ks]);
          st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(hwbf16x8_t, a), __builtin_bit_cast(hwbf16x8_t, qf[ks]),
                                                           st[kb], 0, 0, 0);
        }
      // keys of this lane: 32 kb + 8 (r / 4) + 4 hh + r % 4.  A tile every row of the wave sees completely (below the causal
      // diagonal; image rows) skips the three instructions per score of the bit test -- wave-uniform branch.
      float mloc = -INFINITY;
      if (__ballot(wcur != ~0ull) != 0ull) {
        const uint64_t w2 = wcur >> (hh * 4);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const bool on = (w2 >> (kb * 32 + (r >> 2) * 8 + (r & 3))) & 1ull;
            st[kb][r] = on ? st[kb][r] : -INFINITY;
          }
      }
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, st[kb][r]);
      mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
      const float m_new = fmaxf(m_i, mloc);
      const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
      const float alpha = __builtin_amdgcn_exp2f((m_i - m_use) * c1);      // v_exp_f32: arguments are <= 0, results in [0, 1]
      const float mc = m_use * c1;
      float rs = 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(st[kb][r], c1, -mc)); st[kb][r] = e; rs += e; }
      rs += __shfl_xor(rs, 32, 64);
      l_i = l_i * alpha + rs;
      m_i = m_new;
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[d][r] *= alpha;
      bf16x8_t pf[2][2];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          uint4 u;
          u.x = pack_bf2(st[kb][j * 8 + 0], st[kb][j * 8 + 1]); u.y = pack_bf2(st[kb][j * 8 + 2], st[kb][j * 8 + 3]);
          u.z = pack_bf2(st[kb][j * 8 + 4], st[kb][j * 8 + 5]); u.w = pack_bf2(st[kb][j * 8 + 6], st[kb][j * 8 + 7]);
          pf[kb][j] = __builtin_bit_cast(bf16x8_t, u);
        }
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int d = 0; d < 4; ++d) {                    // four independent accumulators in turn
            const char* vb = Vs + kb * 8192 + j * 4096;
            const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(vb + voff[d][0]));
            const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(vb + voff[d][1]));
            const bf16x8_t a = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            ot[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(hwbf16x8_t, a), __builtin_bit_cast(hwbf16x8_t, pf[kb][j]),
                                                            ot[d], 0, 0, 0);
          }
    }
    t = tn;
    wcur = wnext;
  }
  if (qlive) {
    const float inv = (l_i > 0.f) ? 1.f / l_i : 0.f;
    bf16_t* orow = p.o + ((int64_t)b * p.L + qrow) * p.ldo + h * HD;
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        uint2 w;
        w.x = pack_bf2(ot[d][q * 4 + 0] * inv, ot[d][q * 4 + 1] * inv);
        w.y = pack_bf2(ot[d][q * 4 + 2] * inv, ot[d][q * 4 + 3] * inv);
        *reinterpret_cast<uint2*>(orow + d * 32 + q * 8 + hh * 4) = w;
      }
    if (hh == 0) p.lse[((int64_t)b * p.H + h) * p.L + qrow] = (l_i > 0.f) ? m_i * p.scale + __logf(l_i) : INFINITY;
  }
}

// ================================================================== backward: dQ  (also delta = rowsum(dO * O))
// grid (nQtiles, H, B)
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(AttnArgs p) {
  __shared__ __attribute__((aligned(16))) bf16_t Ks[64 * RM_LD];
  __shared__ __attribute__((aligned(16))) bf16_t Vs[64 * RM_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
  const WgCoord wc = wg_coord((int)blockIdx.x, p.nW, p.H, p.HKV, p.B);
  if (!wc.ok) return;
  const int qt = wc.tile, h = wc.h, b = wc.b;
  const int hk = h / (p.H / p.HKV);
  const int qrow = qt * 64 + wave * 16 + (lane & 15);
  const int qrow_c = min(qrow, p.L - 1);
  const bf16_t* qseq = p.q + (int64_t)b * p.L * p.ldq + h * HD;
  const bf16_t* kseq = p.k + (int64_t)b * p.L * p.ldq + hk * HD;
  const bf16_t* vseq = p.v + (int64_t)b * p.L * p.ldq + hk * HD;
  const bf16_t* doseq = p.dout + (int64_t)b * p.L * p.ldo + h * HD;

  bf16x8_t qf[4], dof[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    qf[ks] = *reinterpret_cast<const bf16x8_t*>(qseq + (int64_t)qrow_c * p.ldq + ks * 32 + g * 8);
    dof[ks] = *reinterpret_cast<const bf16x8_t*>(doseq + (int64_t)qrow_c * p.ldo + ks * 32 + g * 8);
  }
  const float lse = p.lse[((int64_t)b * p.H + h) * p.L + qrow_c];
  // delta = rowsum(dO * O) of this lane's query row, computed here (the four lanes g = 0..3 of a row hold its 128 dims)
  // and published for the dK/dV kernel that runs next: no separate pass over O and dO
  float dl = 0.f;
  {
    const bf16_t* oseq = p.o + (int64_t)b * p.L * p.ldo + h * HD;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8_t of = *reinterpret_cast<const bf16x8_t*>(oseq + (int64_t)qrow_c * p.ldo + ks * 32 + g * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) dl += bf2f((bf16_t)of[e]) * bf2f((bf16_t)dof[ks][e]);
    }
    dl += __shfl_xor(dl, 16, 64);
    dl += __shfl_xor(dl, 32, 64);
    if (g == 0 && qrow < p.L) const_cast<float*>(p.delta)[((int64_t)b * p.H + h) * p.L + qrow] = dl;
  }
  f32x4_t dqt[8];
#pragma unroll
  for (int d = 0; d < 8; ++d) dqt[d] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const float c1 = p.scale * 1.4426950408889634f, lse2 = lse * 1.4426950408889634f, dls = dl * p.scale;
  const uint64_t* wrow = p.bits + ((int64_t)b * p.L + qrow_c) * p.nW;
  const uint8_t* tany = p.tileany + ((int64_t)b * p.nW + qt) * p.nW;

  for (int t = 0; t < p.nW; ++t) {
    if (!tany[t]) continue;
    __syncthreads();
    stage_rm(Ks, kseq, p.ldq, t * 64, p.L, tid);
    stage_rm(Vs, vseq, p.ldq, t * 64, p.L, tid);
    __syncthreads();
    const uint64_t word = wrow[t];
    const uint32_t mlo = (uint32_t)(word >> (g * 4)), mhi = (uint32_t)(word >> (32 + g * 4));   // bit 16 (j & 1) + r
    f32x4_t ds[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4_t s = f32x4_t{0.f, 0.f, 0.f, 0.f}, dp = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rm(Ks, j, ks, lane), qf[ks], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rm(Vs, j, ks, lane), dof[ks], dp, 0, 0, 0);
      }
      // p = exp(scale s - lse) = exp2(c1 s - lse2); dS = p (dP - delta) scale = p fma(dP, scale, -delta scale): two + two
      // instructions per score (the exp() / three-factor form took seven).  (A wave-uniform "tile fully visible" branch around
      // the bit test was measured SLOWER: +9 % on the full mask -- the branch keeps the next block's MFMAs from overlapping.)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool on = ((j < 2 ? mlo : mhi) >> ((j & 1) * 16 + r)) & 1u;
        const float pr = on ? __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], c1, -lse2)) : 0.f;
        ds[j][r] = pr * __builtin_fmaf(dp[r], p.scale, -dls);
      }
    }
    const bf16x8_t sf0 = pack_p_sw(ds[0], ds[1]), sf1 = pack_p_sw(ds[2], ds[3]);
#pragma unroll
    for (int d = 0; d < 8; ++d) dqt[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_trr(Ks, d, 0, lane), sf0, dqt[d], 0, 0, 0);
#pragma unroll
    for (int d = 0; d < 8; ++d) dqt[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_trr(Ks, d, 1, lane), sf1, dqt[d], 0, 0, 0);
  }
  if (qrow < p.L) {
    bf16_t* drow = p.dq + ((int64_t)b * p.L + qrow) * p.ldg + h * HD;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      uint2 w; w.x = pack_bf2(dqt[d][0], dqt[d][1]); w.y = pack_bf2(dqt[d][2], dqt[d][3]);
      *reinterpret_cast<uint2*>(drow + d * 16 + g * 4) = w;
    }
  }
}

// ================================================================== backward: dQ on 32-row waves (round 3)
// The forward's 32-row structure (attn_fwd32_kernel: LDS-DMA ring, one barrier per visible key tile, swizzled tile image) with
// the dQ arithmetic: per key block of 32, S^T = K Q^T and dP^T = V dO^T (two independent accumulator chains), dS = P (dP - delta)
// scale in the exp2 / fma form, then dQ^T += K^T dS^T with K^T fragments read transposed from the SAME staged K tile.  Also
// publishes delta = rowsum(dO * O) for the dK / dV kernel that runs next.
template <int NW>
__global__ __launch_bounds__(64 * NW, 2) void attn_bwd_dq32_kernel(AttnArgs p) {
  __shared__ __attribute__((aligned(16))) char ring[2 * 2 * T32_BYTES];       // [slot][K | V]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 31, hh = lane >> 5, i16 = lane & 15, grp = (lane >> 4) & 1;
  constexpr int ROWS = 32 * NW;
  const WgCoord wc = wg_coord((int)blockIdx.x, (p.L + ROWS - 1) / ROWS, p.H, p.HKV, p.B, true, p.order != 0);
  if (!wc.ok) return;
  const int qt = wc.tile, h = wc.h, b = wc.b;
  const int hk = h / (p.H / p.HKV);
  const int row0 = p.L - ((p.L + ROWS - 1) / ROWS) * ROWS + qt * ROWS;       // end-aligned tiles: < 0 in the first one
  const int qrow = row0 + wave * 32 + n;
  const bool qlive = qrow >= 0;                                              // (always < L)
  const int qrow_c = max(qrow, 0);
  const bf16_t* qseq = p.q + (int64_t)b * p.L * p.ldq + h * HD;
  const bf16_t* kseq = p.k + (int64_t)b * p.L * p.ldq + hk * HD;
  const bf16_t* vseq = p.v + (int64_t)b * p.L * p.ldq + hk * HD;
  const bf16_t* doseq = p.dout + (int64_t)b * p.L * p.ldo + h * HD;
  const bf16_t* oseq = p.o + (int64_t)b * p.L * p.ldo + h * HD;

  bf16x8_t qf[8], dof[8];
  float dl = 0.f;
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    qf[ks] = *reinterpret_cast<const bf16x8_t*>(qseq + (int64_t)qrow_c * p.ldq + ks * 16 + hh * 8);
    dof[ks] = *reinterpret_cast<const bf16x8_t*>(doseq + (int64_t)qrow_c * p.ldo + ks * 16 + hh * 8);
    const bf16x8_t of = *reinterpret_cast<const bf16x8_t*>(oseq + (int64_t)qrow_c * p.ldo + ks * 16 + hh * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) dl += bf2f((bf16_t)of[e]) * bf2f((bf16_t)dof[ks][e]);
  }
  dl += __shfl_xor(dl, 32, 64);                           // the two lanes of a row hold 64 of its 128 dims each
  if (hh == 0 && qlive) const_cast<float*>(p.delta)[((int64_t)b * p.H + h) * p.L + qrow] = dl;
  const float lse = p.lse[((int64_t)b * p.H + h) * p.L + qrow_c];
  const float c1 = p.scale * 1.4426950408889634f, lse2 = lse * 1.4426950408889634f, dls = dl * p.scale;
  f32x16_t dqt[4];
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) dqt[d][r] = 0.f;
  const uint64_t* wrow = p.bits + ((int64_t)b * p.L + qrow_c) * p.nW;
  uint64_t vis, minem;
  {
    bool v = false, m = false;
    if (lane < p.nW) {
      v = rows_see_tile(p.tileany, b, p.nW, row0, row0 + ROWS - 1, p.L, lane);
      m = rows_see_tile(p.tileany, b, p.nW, row0 + wave * 32, row0 + wave * 32 + 31, p.L, lane);
    }
    vis = __ballot(v);
    minem = __ballot(m);
  }
  auto next_visible = [&](int t) {
    const uint64_t rest = t < 64 ? vis >> t : 0ull;
    return rest ? t + __builtin_ctzll(rest) : p.nW;
  };
  // byte offset of this lane's 16 bytes inside a 64-key tile for each of its DMA instructions: computed ONCE -- recomputed per tile
  // (clamp, 64-bit row x stride product, swizzle) it was 43 VALU + ~40 SALU instructions = a sixth of the loop's issue slots
  uint32_t toff[16 / NW];
#pragma unroll
  for (int i = 0; i < 16 / NW; ++i) {
    const int row = (i * NW + wave) * 4 + (lane >> 4);
    toff[i] = (uint32_t)(row * p.ldq + (((lane & 15) ^ swz16(row)) << 3)) * 2u;
  }
  auto stage = [&](int t, int slot) {
    char* base = ring + slot * 2 * T32_BYTES;
    const int64_t tb = (int64_t)t * 64 * p.ldq * 2;
    const uint64_t kb = (uint64_t)kseq + tb, vb = (uint64_t)vseq + tb;
    // hand-issued (vmem_asm.h): behind the builtin hipcc put s_waitcnt vmcnt(0) in front of the tile reads below -- the tile
    // requested a moment ago had to LAND before the first MFMA of the current one, i.e. the ring prefetched nothing (round 6)
    if (t * 64 + 64 <= p.L) {
#pragma unroll
      for (int i = 0; i < 16 / NW; ++i) {
        const int inst = i * NW + wave;
        dma16(kb, toff[i], __builtin_amdgcn_readfirstlane(lds_addr_of(base + inst * 1024)));
        dma16(vb, toff[i], __builtin_amdgcn_readfirstlane(lds_addr_of(base + T32_BYTES + inst * 1024)));
      }
    } else {                                             // the sequence's last, partial tile: rows past L repeat row L - 1
#pragma unroll
      for (int i = 0; i < 16 / NW; ++i) {
        const int inst = i * NW + wave;
        const int row = inst * 4 + (lane >> 4);
        const uint32_t off = (uint32_t)((min(t * 64 + row, p.L - 1) - t * 64) * p.ldq + (((lane & 15) ^ swz16(row)) << 3)) * 2u;
        dma16(kb, off, __builtin_amdgcn_readfirstlane(lds_addr_of(base + inst * 1024)));
        dma16(vb, off, __builtin_amdgcn_readfirstlane(lds_addr_of(base + T32_BYTES + inst * 1024)));
      }
    }
  };
  int koff[8], voff[4][2];
  {
    const int sw = swz16(n);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) koff[ks] = n * 256 + (((ks * 2 + hh) ^ sw) << 4);
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int key = 4 * hh + (i16 >> 2) + 8 * e;
        const int chunk = d * 4 + grp * 2 + ((i16 & 3) >> 1);
        voff[d][e] = key * 256 + ((chunk ^ swz16(key)) << 4) + (i16 & 1) * 8;
      }
  }

  int t = next_visible(0);
  if (t < p.nW) stage(t, 0);
  uint64_t wcur = t < p.nW ? wrow[t] : 0ull;             // (first use right behind the loop-top drain: see attn_fwd32_kernel)
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) { tie(qf[ks]); tie(dof[ks]); }
  for (int it = 0; t < p.nW; ++it) {
    const int tn = next_visible(t + 1);
    wait_vm<0>();
    tie(wcur);
    asm volatile("s_barrier" ::: "memory");
    if (tn < p.nW) stage(tn, (it + 1) & 1);
    const uint64_t wnext = tn < p.nW ? wrow[tn] : 0ull;
    if ((minem >> t) & 1ull) {
      const char* Ks = ring + (it & 1) * 2 * T32_BYTES;
      const char* Vs = Ks + T32_BYTES;
      const uint64_t w2 = wcur >> (hh * 4);
      bf16x8_t sf[2][2];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        f32x16_t sc, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sc[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          const bf16x8_t ka = *reinterpret_cast<const bf16x8_t*>(Ks + kb * 8192 + koff[ks]);
          const bf16x8_t va = *reinterpret_cast<const bf16x8_t*>(Vs + kb * 8192 + koff[ks]);
          sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(hwbf16x8_t, ka), __builtin_bit_cast(hwbf16x8_t, qf[ks]), sc, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(hwbf16x8_t, va), __builtin_bit_cast(hwbf16x8_t, dof[ks]), dp, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const bool on = (w2 >> (kb * 32 + (r >> 2) * 8 + (r & 3))) & 1ull;
          const float pr = on ? __builtin_amdgcn_exp2f(__builtin_fmaf(sc[r], c1, -lse2)) : 0.f;
          sc[r] = pr * __builtin_fmaf(dp[r], p.scale, -dls);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          uint4 u;
          u.x = pack_bf2(sc[j * 8 + 0], sc[j * 8 + 1]); u.y = pack_bf2(sc[j * 8 + 2], sc[j * 8 + 3]);
          u.z = pack_bf2(sc[j * 8 + 4], sc[j * 8 + 5]); u.w = pack_bf2(sc[j * 8 + 6], sc[j * 8 + 7]);
          sf[kb][j] = __builtin_bit_cast(bf16x8_t, u);
        }
      }
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            const char* kb_ = Ks + kb * 8192 + j * 4096;
            const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(kb_ + voff[d][0]));
            const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(kb_ + voff[d][1]));
            const bf16x8_t a = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            dqt[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(hwbf16x8_t, a), __builtin_bit_cast(hwbf16x8_t, sf[kb][j]),
                                                             dqt[d], 0, 0, 0);
          }
    }
    t = tn;
    wcur = wnext;
  }
  // register r of block d is column d * 32 + (r >> 2) * 8 + hh * 4 + (r & 3) of this lane's row: the rotary partner (column + 64)
  // is register r of block d + 2 in the same lane
  if (p.rope_cos) {
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = d * 32 + q * 8 + hh * 4;
        const float4 c = *reinterpret_cast<const float4*>(p.rope_cos + (int64_t)qrow_c * (HD / 2) + col);
        const float4 sn = *reinterpret_cast<const float4*>(p.rope_sin + (int64_t)qrow_c * (HD / 2) + col);
        const float cc[4] = {c.x, c.y, c.z, c.w}, ss[4] = {sn.x, sn.y, sn.z, sn.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float g1 = dqt[d][q * 4 + e], g2 = dqt[d + 2][q * 4 + e];
          rope_bwd_pair(g1, g2, cc[e], ss[e]);
          dqt[d][q * 4 + e] = g1; dqt[d + 2][q * 4 + e] = g2;
        }
      }
  }
  if (qlive) {
    bf16_t* drow = p.dq + ((int64_t)b * p.L + qrow) * p.ldg + h * HD;
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        uint2 w;
        w.x = pack_bf2(dqt[d][q * 4 + 0], dqt[d][q * 4 + 1]);
        w.y = pack_bf2(dqt[d][q * 4 + 2], dqt[d][q * 4 + 3]);
        *reinterpret_cast<uint2*>(drow + d * 32 + q * 8 + hh * 4) = w;
      }
  }
  if (p.dbias) {
    // bias gradient of the q projection: column sums of the stored (bf16) values over this workgroup's rows
    float x[64];
    const bool live = qlive;
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) x[d * 16 + r] = live ? bf2f(f2bf(dqt[d][r])) : 0.f;
    lanes32_colsum(x, lane);
    __syncthreads();                                             // every wave is done with the K / V ring
    float* red = reinterpret_cast<float*>(ring);                 // [NW][128]
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int v = 2 * n + e, d = v >> 4, r = v & 15;
      red[wave * HD + d * 32 + (r >> 2) * 8 + hh * 4 + (r & 3)] = x[e];
    }
    __syncthreads();
    if (tid < HD) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) t += red[w * HD + tid];
      atomicAdd(p.dbias + h * HD + tid, t);
    }
  }
}

// ================================================================== backward: dK, dV
// grid (nKVtiles, HKV, B); each wave owns 16 keys (a lane owns key column kv = lane&15), walks the
// H/HKV query heads of its group and all query tiles; dK^T/dV^T accumulate in registers, no atomics.
// SPLIT: grid (nKVtiles, H, B) -- one query head per workgroup, 6x the workgroups (a causal mask makes key tile 0
// thirteen times the work of the last one, and nKV*HKV*B = 416 workgroups do not even fill the 512 slots once);
// partial dK/dV go to an fp32 workspace with row-contiguous atomics (transposed through LDS), finished by
// dkv_finish_kernel.
template <bool SPLIT>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(AttnArgs p) {
  __shared__ __attribute__((aligned(16))) bf16_t QD[2 * 64 * RM_LD];  // Q rows | dO rows (one array: reused as the
  bf16_t* Qs = QD;                                                    // SPLIT epilogue's fp32 transpose scratch)
  bf16_t* Ds = QD + 64 * RM_LD;
  __shared__ __attribute__((aligned(16))) float lse_s[64], dl_s[64];       // lse log2(e) | delta scale of the staged query rows
  __shared__ uint64_t word_s[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
  const int grp = p.H / p.HKV;
  const WgCoord wc = wg_coord((int)blockIdx.x, p.nW, SPLIT ? p.H : p.HKV, p.HKV, p.B);       // (not SPLIT: one workgroup per kv head)
  if (!wc.ok) return;
  const int t = wc.tile, b = wc.b;
  const int hk = SPLIT ? wc.h / grp : wc.h;
  const int krow = t * 64 + wave * 16 + (lane & 15);
  const int krow_c = min(krow, p.L - 1);
  const bf16_t* kseq = p.k + (int64_t)b * p.L * p.ldq + hk * HD;
  const bf16_t* vseq = p.v + (int64_t)b * p.L * p.ldq + hk * HD;
  bf16x8_t kf[4], vf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    kf[ks] = *reinterpret_cast<const bf16x8_t*>(kseq + (int64_t)krow_c * p.ldq + ks * 32 + g * 8);
    vf[ks] = *reinterpret_cast<const bf16x8_t*>(vseq + (int64_t)krow_c * p.ldq + ks * 32 + g * 8);
  }
  f32x4_t dkt[8], dvt[8];
#pragma unroll
  for (int d = 0; d < 8; ++d) { dkt[d] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dvt[d] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
  const int kbit = wave * 16 + (lane & 15);
  const float c1 = p.scale * 1.4426950408889634f;

  for (int hh = SPLIT ? wc.h % grp : 0; hh < (SPLIT ? wc.h % grp + 1 : grp); ++hh) {
    const int h = hk * grp + hh;
    const bf16_t* qseq = p.q + (int64_t)b * p.L * p.ldq + h * HD;
    const bf16_t* doseq = p.dout + (int64_t)b * p.L * p.ldo + h * HD;
    for (int qt = 0; qt < p.nW; ++qt) {
      if (!p.tileany[((int64_t)b * p.nW + qt) * p.nW + t]) continue;
      __syncthreads();
      stage_rm(Qs, qseq, p.ldq, qt * 64, p.L, tid);
      stage_rm(Ds, doseq, p.ldo, qt * 64, p.L, tid);      // query rows past L repeat the last row: their mask words are 0
      if (tid < 64) {
        const int qr = qt * 64 + tid;
        const bool ok = qr < p.L;
        const int qc = min(qr, p.L - 1);
        lse_s[tid] = p.lse[((int64_t)b * p.H + h) * p.L + qc] * 1.4426950408889634f;       // exp2 domain
        dl_s[tid] = p.delta[((int64_t)b * p.H + h) * p.L + qc] * p.scale;                  // delta scale
        word_s[tid] = ok ? p.bits[((int64_t)b * p.L + qc) * p.nW + t] : 0ull;              // rows past L contribute nothing
      }
      __syncthreads();
      f32x4_t pr[4], ds[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {       // query block j: rows j*16 + g*4 + r, column = this lane's key
        f32x4_t s = f32x4_t{0.f, 0.f, 0.f, 0.f}, dp = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rm(Qs, j, ks, lane), kf[ks], s, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rm(Ds, j, ks, lane), vf[ks], dp, 0, 0, 0);
        }
        const f32x4_t ls = *reinterpret_cast<const f32x4_t*>(lse_s + j * 16 + g * 4);
        const f32x4_t dl4 = *reinterpret_cast<const f32x4_t*>(dl_s + j * 16 + g * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ql = j * 16 + g * 4 + r;
          const bool on = (word_s[ql] >> kbit) & 1ull;
          const float e = on ? __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], c1, -ls[r])) : 0.f;
          pr[j][r] = e;
          ds[j][r] = e * __builtin_fmaf(dp[r], p.scale, -dl4[r]);
        }
      }
      const bf16x8_t pf0 = pack_p(pr[0], pr[1]), pf1 = pack_p(pr[2], pr[3]);
      const bf16x8_t sf0 = pack_p(ds[0], ds[1]), sf1 = pack_p(ds[2], ds[3]);
#pragma unroll
      for (int d = 0; d < 8; ++d) {                         // (16 independent accumulators per pass: no MFMA waits for its predecessor)
        dvt[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_trr(Ds, d, 0, lane), pf0, dvt[d], 0, 0, 0);
        dkt[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_trr(Qs, d, 0, lane), sf0, dkt[d], 0, 0, 0);
      }
#pragma unroll
      for (int d = 0; d < 8; ++d) {
        dvt[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_trr(Ds, d, 1, lane), pf1, dvt[d], 0, 0, 0);
        dkt[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_trr(Qs, d, 1, lane), sf1, dkt[d], 0, 0, 0);
      }
    }
  }
  if constexpr (SPLIT) {
    // wave-private transpose through LDS (the staging tiles are dead): [16 keys][128 d] fp32 per tensor, then every
    // atomic instruction covers 64 consecutive floats of one key row
    __syncthreads();
    static_assert(4 * 16 * 132 * 4 <= 2 * 64 * RM_LD * 2, "transpose scratch must fit in QD");
    float* tw = reinterpret_cast<float*>(QD) + wave * (16 * 132);
    const int ldws = 2 * p.HKV * HD;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
      for (int d = 0; d < 8; ++d)
#pragma unroll
        for (int r = 0; r < 4; ++r) tw[(lane & 15) * 132 + d * 16 + g * 4 + r] = pass == 0 ? dkt[d][r] : dvt[d][r];
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int kr = t * 64 + wave * 16 + i;
        if (kr < p.L) {
          float* dst = p.dkv_ws + ((int64_t)b * p.L + kr) * ldws + (pass * p.HKV + hk) * HD;
          atomicAdd(dst + lane, tw[i * 132 + lane]);
          atomicAdd(dst + 64 + lane, tw[i * 132 + 64 + lane]);
        }
      }
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    return;
  }
  if (krow < p.L) {
    bf16_t* dkrow = p.dk + ((int64_t)b * p.L + krow) * p.ldg + hk * HD;
    bf16_t* dvrow = p.dv + ((int64_t)b * p.L + krow) * p.ldg + hk * HD;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      uint2 w; w.x = pack_bf2(dkt[d][0], dkt[d][1]); w.y = pack_bf2(dkt[d][2], dkt[d][3]);
      *reinterpret_cast<uint2*>(dkrow + d * 16 + g * 4) = w;
      uint2 u; u.x = pack_bf2(dvt[d][0], dvt[d][1]); u.y = pack_bf2(dvt[d][2], dvt[d][3]);
      *reinterpret_cast<uint2*>(dvrow + d * 16 + g * 4) = u;
    }
  }
}

// ------------------------------------------------------------------ dK / dV, split heads, LDS-DMA ring (round 3)
// Same arithmetic and the same 16-key waves as attn_bwd_dkv_kernel<true>; what changes is how a query tile arrives: Q and dO
// rows by LDS-DMA into a two-slot ring one visible tile ahead (swizzled [64 rows][16 chunks] image, chunk ^= swz16b(row): conflict-
// free for the 16-row ds_read_b128 fragments of the 16x16x32 MFMA and for the transposing half-waves), the per-row lse / delta /
// mask words requested one tile ahead into registers of wave 0 and parked in a two-slot LDS record, ONE barrier per tile.  The
// 16-row kernel above stages through registers between two barriers and fetches the row records inside that window.
__device__ __forceinline__ int swz16b(int row) {
  const int m = (row >> 2) & 3;
  return ((row & 3) << 2) | ((0x78 >> (m * 2)) & 3);        // low bits g2[m] = {0, 2, 3, 1}
}
__device__ __forceinline__ bf16x8_t frag16(const char* tile, int rb, int ks, int lane) {
  const int row = rb * 16 + (lane & 15);
  return *reinterpret_cast<const bf16x8_t*>(tile + row * 256 + (((ks * 4 + (lane >> 4)) ^ swz16b(row)) << 4));
}
__device__ __forceinline__ bf16x8_t frag16_tr(const char* tile, int rb, int jp, int lane) {
  const int i16 = lane & 15, g = lane >> 4;
  const int row = jp * 32 + g * 4 + (i16 >> 2);
  const char* p0 = tile + row * 256 + (((rb * 2 + ((i16 & 3) >> 1)) ^ swz16b(row)) << 4) + (i16 & 1) * 8;
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p0);
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p0 + 16 * 256));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_dma_kernel(AttnArgs p) {
  __shared__ __attribute__((aligned(16))) char ring[2 * 2 * T32_BYTES];       // [slot][Q | dO]; reused as the epilogue's fp32 scratch
  __shared__ __attribute__((aligned(16))) float lse_s[2][64], dl_s[2][64];
  __shared__ __attribute__((aligned(16))) uint64_t word_s[2][64];
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = p.H / p.HKV;
  // HPW query heads per workgroup (p.dkv_heads: 1 or 2, dividing the group): the workgroup walks its heads one after the other
  // with the SAME dK / dV accumulators, so the fp32 atomics into the split-head workspace (64 KB per workgroup, 160 MB per launch
  // at one head each) and the finishing pass's contention halve at two
  const int HPW = p.dkv_heads;
  const WgCoord wc = wg_coord((int)blockIdx.x, p.nW, p.H / HPW, p.HKV, p.B, false, p.order != 0);
  if (!wc.ok) return;
  const int t = wc.tile, b = wc.b, h0 = wc.h * HPW, hk = h0 / grp;
  const int krow = t * 64 + wave * 16 + (lane & 15);
  const int krow_c = min(krow, p.L - 1);
  const bf16_t* kseq = p.k + (int64_t)b * p.L * p.ldq + hk * HD;
  const bf16_t* vseq = p.v + (int64_t)b * p.L * p.ldq + hk * HD;
  const bf16_t* qseq = p.q + (int64_t)b * p.L * p.ldq + h0 * HD;
  const bf16_t* doseq = p.dout + (int64_t)b * p.L * p.ldo + h0 * HD;
  int h = h0;
  bf16x8_t kf[4], vf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    kf[ks] = *reinterpret_cast<const bf16x8_t*>(kseq + (int64_t)krow_c * p.ldq + ks * 32 + g * 8);
    vf[ks] = *reinterpret_cast<const bf16x8_t*>(vseq + (int64_t)krow_c * p.ldq + ks * 32 + g * 8);
  }
  f32x4_t dkt[8], dvt[8];
#pragma unroll
  for (int d = 0; d < 8; ++d) { dkt[d] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dvt[d] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
  const int kbit = wave * 16 + (lane & 15);
  const float c1 = p.scale * 1.4426950408889634f;
  // visible query tiles of this key tile (nW <= 64), fetched once
  uint64_t vis;
  {
    bool v = false;
    if (lane < p.nW) v = p.tileany[((int64_t)b * p.nW + lane) * p.nW + t];
    vis = __ballot(v);
  }
  auto next_visible = [&](int q) {
    const uint64_t rest = q < 64 ? vis >> q : 0ull;
    return rest ? q + __builtin_ctzll(rest) : p.nW;
  };
  // lane offsets inside a query tile (Q rows and dO rows have their own strides), computed once (see attn_fwd32_kernel)
  uint32_t qoff[4], dooff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (i * 4 + wave) * 4 + (lane >> 4);
    const int c8 = ((lane & 15) ^ swz16b(row)) << 3;
    qoff[i] = (uint32_t)(row * p.ldq + c8) * 2u;
    dooff[i] = (uint32_t)(row * p.ldo + c8) * 2u;
  }
  auto stage = [&](int qt, int slot) {
    char* base = ring + slot * 2 * T32_BYTES;
    const uint64_t qb = (uint64_t)qseq + (int64_t)qt * 64 * p.ldq * 2, dob = (uint64_t)doseq + (int64_t)qt * 64 * p.ldo * 2;
    if (qt * 64 + 64 <= p.L) {                           // (hand-issued: see attn_fwd32_kernel)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int inst = i * 4 + wave;
        dma16(qb, qoff[i], __builtin_amdgcn_readfirstlane(lds_addr_of(base + inst * 1024)));
        dma16(dob, dooff[i], __builtin_amdgcn_readfirstlane(lds_addr_of(base + T32_BYTES + inst * 1024)));
      }
    } else {                                             // rows past L repeat the last row: their mask words are 0
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int inst = i * 4 + wave;
        const int row = inst * 4 + (lane >> 4);
        const int c8 = ((lane & 15) ^ swz16b(row)) << 3;
        const int rr = min(qt * 64 + row, p.L - 1) - qt * 64;
        dma16(qb, (uint32_t)(rr * p.ldq + c8) * 2u, __builtin_amdgcn_readfirstlane(lds_addr_of(base + inst * 1024)));
        dma16(dob, (uint32_t)(rr * p.ldo + c8) * 2u, __builtin_amdgcn_readfirstlane(lds_addr_of(base + T32_BYTES + inst * 1024)));
      }
    }
  };
  // row record of a query tile (wave 0, one lane per row): requested a tile ahead, parked in LDS before the tile's barrier
  float r_lse = 0.f, r_dl = 0.f;
  uint64_t r_word = 0ull;
  bool r_in = false;
  auto fetch_rows = [&](int qt) {                        // RAW values: their first use (tie) sits behind the next step's drain
    const int qr = qt * 64 + lane;
    const int qc = min(qr, p.L - 1);
    r_lse = p.lse[((int64_t)b * p.H + h) * p.L + qc];
    r_dl = p.delta[((int64_t)b * p.H + h) * p.L + qc];
    r_word = p.bits[((int64_t)b * p.L + qc) * p.nW + t];   // (unconditional: a branch here makes hipcc guard the register with a wait
    r_in = qr < p.L;                                      //  that drains the DMA just issued); rows past L contribute nothing
  };

  // the (head, visible query tile) sequence: head h0's tiles, then head h0 + 1's, ... -- one ring, one barrier per step
  int qt = next_visible(0);
  int hleft = qt < p.nW ? HPW - 1 : 0;                    // heads still to come after the current one
  if (qt < p.nW) { stage(qt, 0); if (wave == 0) fetch_rows(qt); }
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) { tie(kf[ks]); tie(vf[ks]); }          // (the key / value fragments' own wait falls here, ahead of the loop)
  for (int it = 0; qt < p.nW; ++it) {
    int qn = next_visible(qt + 1);
    const int slot = it & 1;
    wait_vm<0>();                                        // this wave's share of the tile has landed (and wave 0's row records)
    if (wave == 0) {
      tie(r_lse); tie(r_dl); tie(r_word);
      lse_s[slot][lane] = r_lse * 1.4426950408889634f;     // exp2 domain
      dl_s[slot][lane] = r_dl * p.scale;                   // delta scale
      word_s[slot][lane] = r_in ? r_word : 0ull;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    if (qn >= p.nW && hleft > 0) {                        // next head: its Q / dO / row records from the first visible tile on
      --hleft; ++h;
      qseq += HD; doseq += HD;
      qn = next_visible(0);
    }
    if (qn < p.nW) { stage(qn, slot ^ 1); if (wave == 0) fetch_rows(qn); }
    const char* Qs = ring + slot * 2 * T32_BYTES;
    const char* Ds = Qs + T32_BYTES;
    f32x4_t pr[4], ds[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {       // query block j: rows j*16 + g*4 + r, column = this lane's key
      f32x4_t sc = f32x4_t{0.f, 0.f, 0.f, 0.f}, dp = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag16(Qs, j, ks, lane), kf[ks], sc, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag16(Ds, j, ks, lane), vf[ks], dp, 0, 0, 0);
      }
      const f32x4_t ls = *reinterpret_cast<const f32x4_t*>(&lse_s[slot][j * 16 + g * 4]);
      const f32x4_t dl4 = *reinterpret_cast<const f32x4_t*>(&dl_s[slot][j * 16 + g * 4]);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ql = j * 16 + g * 4 + r;
        const bool on = (word_s[slot][ql] >> kbit) & 1ull;
        const float e = on ? __builtin_amdgcn_exp2f(__builtin_fmaf(sc[r], c1, -ls[r])) : 0.f;
        pr[j][r] = e;
        ds[j][r] = e * __builtin_fmaf(dp[r], p.scale, -dl4[r]);
      }
    }
    const bf16x8_t pf0 = pack_p(pr[0], pr[1]), pf1 = pack_p(pr[2], pr[3]);
    const bf16x8_t sf0 = pack_p(ds[0], ds[1]), sf1 = pack_p(ds[2], ds[3]);
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      dvt[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag16_tr(Ds, d, 0, lane), pf0, dvt[d], 0, 0, 0);
      dkt[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag16_tr(Qs, d, 0, lane), sf0, dkt[d], 0, 0, 0);
    }
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      dvt[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag16_tr(Ds, d, 1, lane), pf1, dvt[d], 0, 0, 0);
      dkt[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag16_tr(Qs, d, 1, lane), sf1, dkt[d], 0, 0, 0);
    }
    qt = qn;
  }
  // wave-private transpose through LDS (the ring is dead once every wave is past its last tile): [16 keys][128 d] fp32 per
  // tensor, then every atomic instruction covers 64 consecutive floats of one key row
  __syncthreads();
  float* tw = reinterpret_cast<float*>(ring) + wave * (16 * 132);
  const int ldws = 2 * p.HKV * HD;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
    for (int d = 0; d < 8; ++d)
#pragma unroll
      for (int r = 0; r < 4; ++r) tw[(lane & 15) * 132 + d * 16 + g * 4 + r] = pass == 0 ? dkt[d][r] : dvt[d][r];
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int kr = t * 64 + wave * 16 + i;
      if (kr < p.L) {
        float* dst = p.dkv_ws + ((int64_t)b * p.L + kr) * ldws + (pass * p.HKV + hk) * HD;
        atomicAdd(dst + lane, tw[i * 132 + lane]);
        atomicAdd(dst + 64 + lane, tw[i * 132 + 64 + lane]);
      }
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

// workspace -> bf16 dK | dV with RoPE transposed on dK and the k | v bias gradient (column sums of the stored values); re-zeroes
// the workspace.  Thread = (row slot, kv head, 4-column group c4 of the first half): it owns columns c4*4.. and their rotary
// partners (+64) of dK and dV for every row of its stripe, so the column sums stay in registers until the end.
__global__ __launch_bounds__(256) void dkv_finish_rope_kernel(float* __restrict__ ws, bf16_t* __restrict__ dk, bf16_t* __restrict__ dv,
                                                              int64_t ldg, int64_t tokens, int L, int HKV,
                                                              const float* __restrict__ cs, const float* __restrict__ sn,
                                                              float* __restrict__ dbias_k, float* __restrict__ dbias_v,
                                                              int rows_per_block) {
  __shared__ float red[256 * 16];
  const int per_row = HKV * 16;                              // threads per row (HKV * 16 <= 256, a power of two)
  const int slot = threadIdx.x / per_row, nslots = 256 / per_row;
  const int hk = (threadIdx.x % per_row) >> 4, c4 = threadIdx.x & 15;
  const int kvw = HKV * HD;
  float sum[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) sum[i] = 0.f;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(tokens, r0 + rows_per_block);
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  // two rows per thread in flight (the pass is latency-bound at the ~128 workgroups the atomics at its end allow)
  for (int64_t rowa = r0 + slot; rowa < r1; rowa += 2 * nslots) {
    float4 k1[2], k2[2], v1[2], v2[2];
    bool ok[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int64_t row = rowa + u * nslots;
      ok[u] = row < r1;
      float* wk = ws + (ok[u] ? row : rowa) * 2 * kvw + hk * HD + c4 * 4;
      float* wv = wk + kvw;
      k1[u] = *reinterpret_cast<float4*>(wk); k2[u] = *reinterpret_cast<float4*>(wk + HD / 2);
      v1[u] = *reinterpret_cast<float4*>(wv); v2[u] = *reinterpret_cast<float4*>(wv + HD / 2);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (!ok[u]) continue;
      const int64_t row = rowa + u * nslots;
      float* wk = ws + row * 2 * kvw + hk * HD + c4 * 4;
      float* wv = wk + kvw;
      *reinterpret_cast<float4*>(wk) = z; *reinterpret_cast<float4*>(wk + HD / 2) = z;
      *reinterpret_cast<float4*>(wv) = z; *reinterpret_cast<float4*>(wv + HD / 2) = z;
      if (cs) {
        const int pos = (int)(row % L);
        const float4 c = *reinterpret_cast<const float4*>(cs + (int64_t)pos * (HD / 2) + c4 * 4);
        const float4 s = *reinterpret_cast<const float4*>(sn + (int64_t)pos * (HD / 2) + c4 * 4);
        rope_bwd_pair(k1[u].x, k2[u].x, c.x, s.x); rope_bwd_pair(k1[u].y, k2[u].y, c.y, s.y);
        rope_bwd_pair(k1[u].z, k2[u].z, c.z, s.z); rope_bwd_pair(k1[u].w, k2[u].w, c.w, s.w);
      }
      uint2 a, bq, cq, dq;
      a.x = pack_bf2(k1[u].x, k1[u].y); a.y = pack_bf2(k1[u].z, k1[u].w);
      bq.x = pack_bf2(k2[u].x, k2[u].y); bq.y = pack_bf2(k2[u].z, k2[u].w);
      cq.x = pack_bf2(v1[u].x, v1[u].y); cq.y = pack_bf2(v1[u].z, v1[u].w);
      dq.x = pack_bf2(v2[u].x, v2[u].y); dq.y = pack_bf2(v2[u].z, v2[u].w);
      bf16_t* okp = dk + row * ldg + hk * HD + c4 * 4;
      bf16_t* ovp = dv + row * ldg + hk * HD + c4 * 4;
      *reinterpret_cast<uint2*>(okp) = a; *reinterpret_cast<uint2*>(okp + HD / 2) = bq;
      *reinterpret_cast<uint2*>(ovp) = cq; *reinterpret_cast<uint2*>(ovp + HD / 2) = dq;
      const uint32_t pk[8] = {a.x, a.y, bq.x, bq.y, cq.x, cq.y, dq.x, dq.y};
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        sum[2 * i] += bf2f((bf16_t)(pk[i] & 0xffffu));
        sum[2 * i + 1] += bf2f((bf16_t)(pk[i] >> 16));
      }
    }
  }
  if (!dbias_k) return;
#pragma unroll
  for (int i = 0; i < 16; ++i) red[i * 256 + threadIdx.x] = sum[i];
  __syncthreads();
  // 16 values x per_row column owners, summed over the row slots: value i of owner o -> column (i: 0-3 k lo, 4-7 k hi, 8-11 v lo, 12-15 v hi)
  for (int e = threadIdx.x; e < 16 * per_row; e += 256) {
    const int i = e / per_row, o = e % per_row;
    float t = 0.f;
    for (int sl = 0; sl < nslots; ++sl) t += red[i * 256 + sl * per_row + o];
    const int ohk = o >> 4, oc4 = o & 15;
    const int col = ohk * HD + ((i >> 2) & 1) * (HD / 2) + oc4 * 4 + (i & 3);
    atomicAdd((i < 8 ? dbias_k : dbias_v) + col, t);
  }
}

// workspace -> bf16 dK | dV (row stride ldg) and re-zero the workspace for the next layer
__global__ __launch_bounds__(256) void dkv_finish_kernel(float* __restrict__ ws, bf16_t* __restrict__ dk, bf16_t* __restrict__ dv,
                                                         int64_t ldg, int64_t tokens, int kvw) {
  const int per_row = 2 * kvw / 4;
  const int64_t total = tokens * per_row;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = idx / per_row; const int c = (int)(idx % per_row) * 4;
    float4* src = reinterpret_cast<float4*>(ws + row * 2 * kvw + c);
    const float4 v = *src;
    *src = make_float4(0.f, 0.f, 0.f, 0.f);
    uint2 w; w.x = pack_bf2(v.x, v.y); w.y = pack_bf2(v.z, v.w);
    bf16_t* dst = c < kvw ? dk + row * ldg + c : dv + row * ldg + (c - kvw);
    *reinterpret_cast<uint2*>(dst) = w;
  }
}

int check_common(const char* fn, int64_t B, int64_t L, int64_t Lp, int H, int HKV, int hd, int64_t ldq) {
  UG_REQUIRE(B > 0 && L > 0, "%s: empty batch", fn);
  UG_REQUIRE(hd == HD, "%s: head_dim %d unsupported (this build is specialised for 128)", fn, hd);
  UG_REQUIRE(H % HKV == 0, "%s: H=%d not a multiple of HKV=%d", fn, H, HKV);
  UG_REQUIRE(Lp % 64 == 0 && Lp >= L, "%s: Lp=%ld must be a multiple of 64 and >= L", fn, (long)Lp);
  UG_REQUIRE(ldq % 8 == 0, "%s: row stride must be a multiple of 8 elements", fn);
  return UG_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------ C ABI
extern "C" int ug_attn_mask_compress(const void* mask, int mask_dtype, int64_t stride_b, int64_t stride_row,
                                     uint64_t* bits, uint8_t* tileany, int64_t B, int64_t L, int* err_flag,
                                     hipStream_t st) {
  UG_REQUIRE(B > 0 && L > 0 && mask && bits && tileany, "ug_attn_mask_compress: bad args");
  const int nW = (int)((L + 63) / 64);
  const int64_t nwaves = B * L * nW;
  dim3 grid((unsigned)((nwaves + 3) / 4)), block(256);
  switch (mask_dtype) {
    case UG_MASK_F32: hipLaunchKernelGGL(mask_compress_kernel<float>, grid, block, 0, st, (const float*)mask, stride_b, stride_row, bits, (int)B, (int)L, nW, err_flag); break;
    case UG_MASK_BF16: hipLaunchKernelGGL(mask_compress_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)mask, stride_b, stride_row, bits, (int)B, (int)L, nW, err_flag); break;
    case UG_MASK_I64: hipLaunchKernelGGL(mask_compress_kernel<int64_t>, grid, block, 0, st, (const int64_t*)mask, stride_b, stride_row, bits, (int)B, (int)L, nW, err_flag); break;
    case UG_MASK_BOOL: hipLaunchKernelGGL(mask_compress_kernel<uint8_t>, grid, block, 0, st, (const uint8_t*)mask, stride_b, stride_row, bits, (int)B, (int)L, nW, err_flag); break;
    default: ug_set_error("ug_attn_mask_compress: unknown mask dtype %d", mask_dtype); return UG_ERR_ARG;
  }
  UG_CHECK_LAUNCH("ug_attn_mask_compress");
  const int64_t ntile = B * nW * nW;
  hipLaunchKernelGGL(mask_tiles_kernel, dim3((unsigned)((ntile + 3) / 4)), block, 0, st, bits, tileany, (int)B, (int)L, nW);
  UG_CHECK_LAUNCH("ug_attn_mask_compress(tiles)");
  return UG_OK;
}

extern "C" int ug_attn_mask_causal(const uint8_t* key_valid, uint64_t* bits, uint8_t* tileany, int64_t B, int64_t L,
                                   hipStream_t st) {
  UG_REQUIRE(B > 0 && L > 0 && bits && tileany, "ug_attn_mask_causal: bad args");
  const int nW = (int)((L + 63) / 64);
  const int64_t nwaves = B * L * nW;
  dim3 block(256);
  hipLaunchKernelGGL(mask_causal_kernel, dim3((unsigned)((nwaves + 3) / 4)), block, 0, st, bits, key_valid, (int)B, (int)L, nW);
  UG_CHECK_LAUNCH("ug_attn_mask_causal");
  const int64_t ntile = B * nW * nW;
  hipLaunchKernelGGL(mask_tiles_kernel, dim3((unsigned)((ntile + 3) / 4)), block, 0, st, bits, tileany, (int)B, (int)L, nW);
  UG_CHECK_LAUNCH("ug_attn_mask_causal(tiles)");
  return UG_OK;
}

extern "C" int ug_attn_mask_from_ids(const int64_t* ids, int64_t B, int64_t L, int64_t pad_id, int64_t soi_id, int64_t eoi_id,
                                     int mode, int* meta_ws, uint8_t* flags_ws, uint64_t* bits, uint8_t* tileany, hipStream_t st) {
  UG_REQUIRE(ids && bits && tileany && meta_ws && flags_ws && B > 0 && L > 0 && mode >= 0 && mode <= 2,
             "ug_attn_mask_from_ids: bad args");
  const int nW = (int)((L + 63) / 64);
  hipLaunchKernelGGL(mask_ids_meta_kernel, dim3((unsigned)B), dim3(64), 0, st, ids, (int)L, pad_id, soi_id, eoi_id, meta_ws, flags_ws);
  UG_CHECK_LAUNCH("ug_attn_mask_from_ids(meta)");
  const int64_t nwaves = B * L * nW;
  hipLaunchKernelGGL(mask_ids_kernel, dim3((unsigned)((nwaves + 3) / 4)), dim3(256), 0, st, bits, meta_ws, flags_ws, (int)B, (int)L, nW, mode);
  UG_CHECK_LAUNCH("ug_attn_mask_from_ids");
  const int64_t ntile = B * nW * nW;
  hipLaunchKernelGGL(mask_tiles_kernel, dim3((unsigned)((ntile + 3) / 4)), dim3(256), 0, st, bits, tileany, (int)B, (int)L, nW);
  UG_CHECK_LAUNCH("ug_attn_mask_from_ids(tiles)");
  return UG_OK;
}

extern "C" int ug_attn_fwd(const void* q, const void* k, const void* v, int64_t ldq, void* o,
                           int64_t ldo, float* lse, const uint64_t* bits, const uint8_t* tileany, int64_t B, int64_t L,
                           int64_t Lp, int H, int HKV, int head_dim, float scale, hipStream_t st) {
  if (int rc = check_common("ug_attn_fwd", B, L, Lp, H, HKV, head_dim, ldq)) return rc;
  UG_REQUIRE(ug_aligned16(q) && ug_aligned16(k) && ug_aligned16(v) && ((uintptr_t)o & 7) == 0 && ldo % 4 == 0,
             "ug_attn_fwd: alignment");
  AttnArgs a{};
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v;
  a.o = (bf16_t*)o; a.lse = lse; a.bits = bits; a.tileany = tileany;
  a.ldq = ldq; a.ldo = ldo; a.B = (int)B; a.L = (int)L; a.Lp = (int)Lp; a.nW = (int)((L + 63) / 64);
  a.H = H; a.HKV = HKV; a.scale = scale;
  { static const int abl = [] { const char* e = getenv("UNIGEN_ATTN_ABLATE"); return e ? atoi(e) : 0; }(); a.ablate = abl; }
  // 128-row query tiles (eight waves share each staged K / V tile) once there are enough of them to fill the chip
  static const int use32 = [] { const char* e = getenv("UNIGEN_ATTN_FWD32"); return e ? atoi(e) : 1; }();
  { static const int ord = [] { const char* e = getenv("UNIGEN_ATTN_ORDER"); return e ? atoi(e) : 1; }(); a.order = ord & 1; }
  if (use32 && L >= 256 && L <= 4096 && (int64_t)((L + 127) / 128) * H * B >= 512)
    hipLaunchKernelGGL(attn_fwd32_kernel<4>, dim3(wg_grid((L + 127) / 128, H, HKV, B)), dim3(256), 0, st, a);
  else if (L >= 256 && (int64_t)((L + 127) / 128) * H * B >= 512)
    hipLaunchKernelGGL(attn_fwd_kernel<8>, dim3(wg_grid((L + 127) / 128, H, HKV, B)), dim3(512), 0, st, a);
  else
    hipLaunchKernelGGL(attn_fwd_kernel<4>, dim3(wg_grid(a.nW, H, HKV, B)), dim3(256), 0, st, a);
  UG_CHECK_LAUNCH("ug_attn_fwd");
  return UG_OK;
}

extern "C" int ug_attn_bwd(const void* q, const void* k, const void* v, int64_t ldq,
                           const void* o, const void* dout, int64_t ldo, const float* lse,
                           float* delta, void* dq, void* dk, void* dv, int64_t ldg, const uint64_t* bits,
                           const uint8_t* tileany, int64_t B, int64_t L, int64_t Lp, int H, int HKV, int head_dim,
                           float scale, float* dkv_ws, const float* rope_cos, const float* rope_sin, float* dbias, hipStream_t st) {
  if (int rc = check_common("ug_attn_bwd", B, L, Lp, H, HKV, head_dim, ldq)) return rc;
  UG_REQUIRE((rope_cos == nullptr) == (rope_sin == nullptr) && (!rope_cos || (ug_aligned16(rope_cos) && ug_aligned16(rope_sin))),
             "ug_attn_bwd: rope_cos / rope_sin come together, 16-byte aligned");
  UG_REQUIRE(ug_aligned16(dkv_ws), "ug_attn_bwd: workspace alignment");
  UG_REQUIRE(ug_aligned16(q) && ug_aligned16(k) && ug_aligned16(v) && ug_aligned16(dout) && ldo % 8 == 0 && ldg % 4 == 0,
             "ug_attn_bwd: alignment");
  AttnArgs a{};
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v;
  a.dout = (const bf16_t*)dout; a.lse = const_cast<float*>(lse); a.delta = delta;
  a.o = const_cast<bf16_t*>((const bf16_t*)o);
  a.dq = (bf16_t*)dq; a.dk = (bf16_t*)dk; a.dv = (bf16_t*)dv; a.dkv_ws = dkv_ws;
  a.bits = bits; a.tileany = tileany;
  a.ldq = ldq; a.ldo = ldo; a.ldg = ldg; a.B = (int)B; a.L = (int)L; a.Lp = (int)Lp; a.nW = (int)((L + 63) / 64);
  a.H = H; a.HKV = HKV; a.scale = scale;
  { static const int abl = [] { const char* e = getenv("UNIGEN_ATTN_ABLATE"); return e ? atoi(e) : 0; }(); a.ablate = abl; }
  static const int use32 = [] { const char* e = getenv("UNIGEN_ATTN_DQ32"); return e ? atoi(e) : 1; }();
  static const int fuse = [] { const char* e = getenv("UNIGEN_ATTN_BWD_FUSE_ROPE"); return e ? atoi(e) : 1; }();
  const int64_t tokens = B * L;
  // what the fused stores do not cover is done by the stand-alone passes (same arithmetic): ug_rope / ug_colsum_bf16
  auto rope_pass = [&](void* x, int nheads) -> int { return rope_cos ? ug_rope(x, rope_cos, rope_sin, tokens, L, ldg, nheads, HD, 1, st) : UG_OK; };
  auto colsum_pass = [&](const void* x, float* out, int cols) -> int { return dbias ? ug_colsum_bf16(x, ldg, out, tokens, cols, st) : UG_OK; };
  static const int ord = [] { const char* e = getenv("UNIGEN_ATTN_ORDER"); return e ? atoi(e) : 1; }();
  a.order = (ord >> 1) & 1;
  if (use32 && L >= 256 && L <= 4096 && (int64_t)((L + 127) / 128) * H * B >= 512) {
    if (fuse) { a.rope_cos = rope_cos; a.rope_sin = rope_sin; a.dbias = dbias; }
    hipLaunchKernelGGL(attn_bwd_dq32_kernel<4>, dim3(wg_grid((L + 127) / 128, H, HKV, B)), dim3(256), 0, st, a);
    UG_CHECK_LAUNCH("ug_attn_bwd(dq)");
    if (!fuse) { if (int rc = rope_pass(dq, H)) return rc; if (int rc = colsum_pass(dq, dbias, H * HD)) return rc; }
  } else {
    hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3(wg_grid(a.nW, H, HKV, B)), dim3(256), 0, st, a);
    UG_CHECK_LAUNCH("ug_attn_bwd(dq)");
    if (int rc = rope_pass(dq, H)) return rc;
    if (int rc = colsum_pass(dq, dbias, H * HD)) return rc;
  }
  a.rope_cos = a.rope_sin = nullptr; a.dbias = nullptr;
  float* dbias_k = dbias ? dbias + H * HD : nullptr;
  float* dbias_v = dbias ? dbias + (H + HKV) * HD : nullptr;
  if (dkv_ws) {
    static const int dma = [] { const char* e = getenv("UNIGEN_ATTN_DKV_DMA"); return e ? atoi(e) : 1; }();
    a.order = (ord >> 2) & 1;
    static const int hpw_env = [] { const char* e = getenv("UNIGEN_ATTN_DKV_HEADS"); return e ? atoi(e) : 2; }();
    static const int hpw_min_wgs = [] { const char* e = getenv("UNIGEN_ATTN_DKV_MIN_WGS"); return e ? atoi(e) : 1024; }();
    a.dkv_heads = (hpw_env >= 1 && (H / HKV) % hpw_env == 0 && (int64_t)a.nW * (H / hpw_env) * B >= hpw_min_wgs) ? hpw_env : 1;
    if (dma && L <= 4096) hipLaunchKernelGGL(attn_bwd_dkv_dma_kernel, dim3(wg_grid(a.nW, H / a.dkv_heads, HKV, B)), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(attn_bwd_dkv_kernel<true>, dim3(wg_grid(a.nW, H, HKV, B)), dim3(256), 0, st, a);
    UG_CHECK_LAUNCH("ug_attn_bwd(dkv split)");
    if (fuse && (rope_cos || dbias) && HKV <= 16 && (HKV & (HKV - 1)) == 0) {
      // ~128 workgroups: every workgroup ends with one atomic per bias column (512 addresses), and 1 000 same-address atomics
      // in a burst cost more (53 us measured) than the whole streaming pass (9 us)
      static const int fin_wgs = [] { const char* e = getenv("UNIGEN_ATTN_FINISH_WGS"); return e ? atoi(e) : 128; }();
      const int rpb = (int)((tokens + fin_wgs - 1) / fin_wgs < 8 ? 8 : (tokens + fin_wgs - 1) / fin_wgs);
      hipLaunchKernelGGL(dkv_finish_rope_kernel, dim3((unsigned)((tokens + rpb - 1) / rpb)), dim3(256), 0, st, dkv_ws, a.dk, a.dv, ldg,
                         tokens, (int)L, HKV, rope_cos, rope_sin, dbias_k, dbias_v, rpb);
      UG_CHECK_LAUNCH("ug_attn_bwd(dkv finish + rope)");
      return UG_OK;
    }
    const int64_t total = B * L * (2 * HKV * HD / 4);
    int64_t gsz = (total + 255) / 256; if (gsz > 4096) gsz = 4096;
    hipLaunchKernelGGL(dkv_finish_kernel, dim3((unsigned)gsz), dim3(256), 0, st, dkv_ws, a.dk, a.dv, ldg, B * L, HKV * HD);
    UG_CHECK_LAUNCH("ug_attn_bwd(dkv finish)");
  } else {
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<false>, dim3(wg_grid(a.nW, HKV, HKV, B)), dim3(256), 0, st, a);
    UG_CHECK_LAUNCH("ug_attn_bwd(dkv)");
  }
  if (int rc = rope_pass(dk, HKV)) return rc;
  if (int rc = colsum_pass(dk, dbias_k, HKV * HD)) return rc;
  if (int rc = colsum_pass(dv, dbias_v, HKV * HD)) return rc;
  return UG_OK;
}
