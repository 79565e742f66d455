// HBM-bound row/elementwise kernels of the Qwen2.5 backbone for gfx950: RMSNorm fwd/bwd, RoPE,
// SwiGLU, cast, embedding gather / scatter-add, column sums, fused AdamW.
// All are one pass over their operands with 16-byte per-lane accesses and wave64 shuffles for the
// row reductions (no LDS round trip unless a cross-wave sum is needed).
//
// Numerics follow the reference's autocast(bf16) path (SURVEY.md §8a "precision modes", mode A):
// fp32 residual stream and norm statistics, bf16 Linear inputs/outputs, fp32 master weights.
#include <algorithm>
#include "common.h"
#include "unigen_hip.h"

namespace {

// Streaming accesses.  NT bit 0 = non-temporal stores, bit 1 = non-temporal loads; UNIGEN_EW_NT = four hex digits, AdamW |
// rmsnorm_bwd | swiglu_bwd | swiglu_fwd.  Default 0x3032, measured inside the step (tools/probes/ab.sh env UNIGEN_EW_NT ...): the SwiGLU backward
// reads gate | up and d(act) for the last time and nothing reads d(gate | up) before the two GEMMs that follow evict it anyway
// (203 -> ~185 us per launch); the forward's gate | up is next read in the backward, its output at once by the down projection
// (so only its loads); the optimizer touches every byte once per step.  Together 132.8 -> 131.7 ms per step; the RMSNorm
// backward did not move (0x0232 vs 0x0032 within noise).
typedef unsigned int ntu4_t __attribute__((ext_vector_type(4)));
typedef unsigned int ntu2_t __attribute__((ext_vector_type(2)));
template <int NT, typename T>
__device__ __forceinline__ T ld_stream(const T* p) {
  static_assert(sizeof(T) == 16 || sizeof(T) == 8, "ld_stream: 8- or 16-byte accesses");
  if constexpr ((NT & 2) && sizeof(T) == 16) return __builtin_bit_cast(T, __builtin_nontemporal_load(reinterpret_cast<const ntu4_t*>(p)));
  else if constexpr ((NT & 2) && sizeof(T) == 8) return __builtin_bit_cast(T, __builtin_nontemporal_load(reinterpret_cast<const ntu2_t*>(p)));
  else return *p;
}
template <int NT, typename T>
__device__ __forceinline__ void st_stream(T* p, T v) {
  static_assert(sizeof(T) == 16 || sizeof(T) == 8, "st_stream: 8- or 16-byte accesses");
  if constexpr ((NT & 1) && sizeof(T) == 16) __builtin_nontemporal_store(__builtin_bit_cast(ntu4_t, v), reinterpret_cast<ntu4_t*>(p));
  else if constexpr ((NT & 1) && sizeof(T) == 8) __builtin_nontemporal_store(__builtin_bit_cast(ntu2_t, v), reinterpret_cast<ntu2_t*>(p));
  else *p = v;
}
static int ew_nt() { static const int v = [] { const char* e = getenv("UNIGEN_EW_NT"); return e ? (int)strtol(e, nullptr, 16) : 0x3032; }(); return v; }

// =========================================================================== RMSNorm
// y = bf16( w * (x * rsqrt(mean(x^2) + eps)) )      (transformers Qwen2RMSNorm.forward,
// modeling_qwen2.py:246-252 followed by the autocast bf16 cast of the next Linear's input)
// one wave per row; COLS = columns per lane handled in float4 chunks.
template <bool OUT_F32>
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          void* __restrict__ y, float* __restrict__ rstd,
                                                          int rows, int cols, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float4* xr = reinterpret_cast<const float4*>(x + (int64_t)row * cols);
  const int nv = cols >> 2;
  float ss = 0.f;
  for (int i = lane; i < nv; i += 64) {
    const float4 v = xr[i];
    ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  ss = wave_sum(ss);
  const float r = rsqrtf(ss / (float)cols + eps);
  if (lane == 0 && rstd) rstd[row] = r;
  const float4* wr = reinterpret_cast<const float4*>(w);
  for (int i = lane; i < nv; i += 64) {
    const float4 v = xr[i], g = wr[i];
    const float o0 = g.x * (v.x * r), o1 = g.y * (v.y * r), o2 = g.z * (v.z * r), o3 = g.w * (v.w * r);
    if constexpr (OUT_F32) {
      reinterpret_cast<float4*>(reinterpret_cast<float*>(y) + (int64_t)row * cols)[i] = make_float4(o0, o1, o2, o3);
    } else {
      uint2 o; o.x = pack_bf2(o0, o1); o.y = pack_bf2(o2, o3);
      reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(y) + (int64_t)row * cols)[i] = o;
    }
  }
}

// The same for rows of at most 64 * 4 * NV columns with the row kept in registers between the two passes (one read of x instead
// of an HBM read + an L2 re-read), several rows per wave; identical arithmetic and summation order.
template <bool OUT_F32, int NV>
__global__ __launch_bounds__(256) void rmsnorm_fwd_reg_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              void* __restrict__ y, float* __restrict__ rstd,
                                                              int rows, int cols, float eps, int rows_per_wave) {
  const int lane = threadIdx.x & 63;
  const int nv = cols >> 2;
  const float4* wr = reinterpret_cast<const float4*>(w);
  const int r0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * rows_per_wave;
  for (int row = r0; row < min(rows, r0 + rows_per_wave); ++row) {
    const float4* xr = reinterpret_cast<const float4*>(x + (int64_t)row * cols);
    float4 v[NV];
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int i = lane + k * 64;
      if (i < nv) { v[k] = xr[i]; }
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int i = lane + k * 64;
      if (i < nv) ss += v[k].x * v[k].x + v[k].y * v[k].y + v[k].z * v[k].z + v[k].w * v[k].w;
    }
    ss = wave_sum(ss);
    const float r = rsqrtf(ss / (float)cols + eps);
    if (lane == 0 && rstd) rstd[row] = r;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int i = lane + k * 64;
      if (i < nv) {
        const float4 g = wr[i];
        const float o0 = g.x * (v[k].x * r), o1 = g.y * (v[k].y * r), o2 = g.z * (v[k].z * r), o3 = g.w * (v[k].w * r);
        if constexpr (OUT_F32) {
          reinterpret_cast<float4*>(reinterpret_cast<float*>(y) + (int64_t)row * cols)[i] = make_float4(o0, o1, o2, o3);
        } else {
          uint2 o; o.x = pack_bf2(o0, o1); o.y = pack_bf2(o2, o3);
          reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(y) + (int64_t)row * cols)[i] = o;
        }
      }
    }
  }
}

// dres += rstd * (g - xhat * mean(g * xhat)),  g = dy * w,  xhat = x * rstd;   dw += sum_rows dy * xhat
// Each wave walks ROWS_PER_WAVE rows keeping its dw partials in registers (cols <= 64*4*MAXV).
constexpr int RN_MAXV = 8;   // supports cols <= 2048
// NV = float4 slots per lane actually needed (cols <= 256 NV): the 1.5B model's 1536 columns take 6 of the 8 -- 30 fewer live
// registers per lane than the generic form, i.e. more resident waves and more rows in flight for an HBM-latency-bound row walk
template <int NV, int NT>
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const bf16_t* __restrict__ dy, const float* __restrict__ x,
                                                          const float* __restrict__ rstd, const float* __restrict__ w,
                                                          float* __restrict__ dres, float* __restrict__ dw,
                                                          bf16_t* __restrict__ dres_bf16, int rows, int cols,
                                                          int rows_per_block) {
  __shared__ float red[4][64 * 4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int nv = cols >> 2;
  float4 dwp[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) dwp[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4* wr = reinterpret_cast<const float4*>(w);
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(rows, r0 + rows_per_block);
  for (int row = r0 + wave; row < r1; row += 4) {
    const float4* xr = reinterpret_cast<const float4*>(x + (int64_t)row * cols);
    const uint2* dyr = reinterpret_cast<const uint2*>(dy + (int64_t)row * cols);
    float4* dr = reinterpret_cast<float4*>(dres + (int64_t)row * cols);
    const float r = rstd[row];
    float4 g[NV], xh[NV], acc[NV];
    uint2 dyv[NV];
    float dot = 0.f;
    // every load of the row goes out before anything is consumed (x / dy first: the dot product needs them; then the residual
    // gradient this row adds to)
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int i = lane + k * 64;
      if (i < nv) { xh[k] = ld_stream<NT>(xr + i); dyv[k] = ld_stream<NT>(dyr + i); }   // last use of x and dy
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int i = lane + k * 64;
      if (i < nv) acc[k] = dr[i];
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int i = lane + k * 64;
      if (i < nv) {
        const float4 v = xh[k], ww = wr[i]; const uint2 d = dyv[k];
        const float d0 = bf2f(d.x & 0xffff), d1 = bf2f(d.x >> 16), d2 = bf2f(d.y & 0xffff), d3 = bf2f(d.y >> 16);
        xh[k] = make_float4(v.x * r, v.y * r, v.z * r, v.w * r);
        g[k] = make_float4(d0 * ww.x, d1 * ww.y, d2 * ww.z, d3 * ww.w);
        dot += g[k].x * xh[k].x + g[k].y * xh[k].y + g[k].z * xh[k].z + g[k].w * xh[k].w;
        dwp[k].x += d0 * xh[k].x; dwp[k].y += d1 * xh[k].y; dwp[k].z += d2 * xh[k].z; dwp[k].w += d3 * xh[k].w;
      }
    }
    dot = wave_sum(dot) / (float)cols;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int i = lane + k * 64;
      if (i < nv) {
        float4 o = acc[k];
        o.x += r * (g[k].x - xh[k].x * dot); o.y += r * (g[k].y - xh[k].y * dot);
        o.z += r * (g[k].z - xh[k].z * dot); o.w += r * (g[k].w - xh[k].w * dot);
        st_stream<NT>(dr + i, o);
        if (dres_bf16) {          // the next GEMM's bf16 operand, written here instead of by a cast pass over dres
          uint2 ob; ob.x = pack_bf2(o.x, o.y); ob.y = pack_bf2(o.z, o.w);
          reinterpret_cast<uint2*>(dres_bf16 + (int64_t)row * cols)[i] = ob;
        }
      }
    }
  }
  // cross-wave dw reduction through LDS, then one atomic per column per block
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    if (k * 64 >= nv) break;   // uniform
    __syncthreads();
    red[wave][lane * 4 + 0] = dwp[k].x; red[wave][lane * 4 + 1] = dwp[k].y;
    red[wave][lane * 4 + 2] = dwp[k].z; red[wave][lane * 4 + 3] = dwp[k].w;
    __syncthreads();
    const int c = threadIdx.x;           // 256 threads <-> 256 columns of this chunk
    const int col = k * 256 + c;
    if constexpr (!(NT & 4))          // (NT bit 2: timing probe without the dw atomics -- wrong gradients, never a default)
      if (col < cols) atomicAdd(dw + col, red[0][c] + red[1][c] + red[2][c] + red[3][c]);
  }
}

// =========================================================================== RoPE (rotate-half)
// In place on the q and k heads of the fused qkv buffer [tokens, ldq]; cos/sin tables [L, hd/2] fp32
// built on the host exactly as Qwen2RotaryEmbedding does (modeling_qwen2.py:91-102); the products
// and the sum are separate fp32 roundings like apply_rotary_pos_emb (:131-135), then one bf16 round.
template <bool BWD>
__global__ __launch_bounds__(256) void rope_kernel(bf16_t* __restrict__ qkv, const float* __restrict__ cs,
                                                   const float* __restrict__ sn, int64_t tokens, int L, int ldq,
                                                   int nheads, int hd) {
#pragma clang fp contract(off)   // products and sum must round separately (HIP's __fmul_rn is a plain '*')
  const int half = hd >> 1, per_head = half >> 2;          // 4 pairs per thread
  const int64_t total = tokens * nheads * per_head;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int p4 = (int)(idx % per_head);
    const int h = (int)((idx / per_head) % nheads);
    const int64_t t = idx / ((int64_t)per_head * nheads);
    const int pos = (int)(t % L);
    bf16_t* base = qkv + t * ldq + h * hd + p4 * 4;
    const uint2 lo = *reinterpret_cast<const uint2*>(base);
    const uint2 hi = *reinterpret_cast<const uint2*>(base + half);
    const float4 c = *reinterpret_cast<const float4*>(cs + (int64_t)pos * half + p4 * 4);
    const float4 s = *reinterpret_cast<const float4*>(sn + (int64_t)pos * half + p4 * 4);
    const float x1[4] = {bf2f(lo.x & 0xffff), bf2f(lo.x >> 16), bf2f(lo.y & 0xffff), bf2f(lo.y >> 16)};
    const float x2[4] = {bf2f(hi.x & 0xffff), bf2f(hi.x >> 16), bf2f(hi.y & 0xffff), bf2f(hi.y >> 16)};
    const float cc[4] = {c.x, c.y, c.z, c.w}, ss[4] = {s.x, s.y, s.z, s.w};
    float o1[4], o2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      // plain '*' and '+' lexically inside this contract(off) scope: each rounds on its own
      const float a1 = x1[k] * cc[k], a2 = x2[k] * cc[k];
      const float b1 = x2[k] * ss[k], b2 = x1[k] * ss[k];
      if constexpr (!BWD) { o1[k] = a1 - b1; o2[k] = a2 + b2; }
      else { o1[k] = a1 + b1; o2[k] = a2 - b2; }      // transpose of the rotation
    }
    uint2 olo, ohi;
    olo.x = pack_bf2(o1[0], o1[1]); olo.y = pack_bf2(o1[2], o1[3]);
    ohi.x = pack_bf2(o2[0], o2[1]); ohi.y = pack_bf2(o2[2], o2[3]);
    *reinterpret_cast<uint2*>(base) = olo;
    *reinterpret_cast<uint2*>(base + half) = ohi;
  }
}

// =========================================================================== SwiGLU
// act = bf16( bf16(silu(gate)) * up )     (Qwen2MLP.forward, modeling_qwen2.py:46-48, bf16 autocast:
// silu and the product are separate bf16-rounded ops).  gu = [tokens, 2*I] = [gate | up].
__device__ __forceinline__ float silu_f(float g) { return silu_train(g); }       // (common.h: shared with the GEMM epilogues)

template <int NT>
__global__ __launch_bounds__(256) void swiglu_fwd_kernel(const bf16_t* __restrict__ gu, bf16_t* __restrict__ act,
                                                         int64_t tokens, int I) {
  const int per_row = I >> 3;
  const int64_t total = tokens * per_row;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = idx / per_row; const int c = (int)(idx % per_row) * 8;
    const bf16x8_t g = ld_stream<NT>(reinterpret_cast<const bf16x8_t*>(gu + t * 2 * I + c));
    const bf16x8_t u = ld_stream<NT>(reinterpret_cast<const bf16x8_t*>(gu + t * 2 * I + I + c));
    bf16x8_t o;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float s = bf2f(f2bf(silu_f(bf2f((bf16_t)g[k]))));
      o[k] = (short)f2bf(s * bf2f((bf16_t)u[k]));
    }
    st_stream<NT>(reinterpret_cast<bf16x8_t*>(act + t * I + c), o);
  }
}

// dgate = dact * up * silu'(gate),  dup = dact * silu(gate);   dgu = [dgate | dup]
template <int NT>
__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const bf16_t* __restrict__ gu, const bf16_t* __restrict__ dact,
                                                         bf16_t* __restrict__ dgu, int64_t tokens, int I) {
  const int per_row = I >> 3;
  const int64_t total = tokens * per_row;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = idx / per_row; const int c = (int)(idx % per_row) * 8;
    const bf16x8_t g = ld_stream<NT>(reinterpret_cast<const bf16x8_t*>(gu + t * 2 * I + c));        // last use of gate | up
    const bf16x8_t u = ld_stream<NT>(reinterpret_cast<const bf16x8_t*>(gu + t * 2 * I + I + c));
    const bf16x8_t d = ld_stream<NT>(reinterpret_cast<const bf16x8_t*>(dact + t * I + c));
    bf16x8_t og, ou;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float dg_, du_;
      swiglu_bwd_elem(bf2f((bf16_t)g[k]), bf2f((bf16_t)u[k]), bf2f((bf16_t)d[k]), dg_, du_);      // (common.h: shared with the dgrad epilogue)
      ou[k] = (short)f2bf(du_);
      og[k] = (short)f2bf(dg_);
    }
    st_stream<NT>(reinterpret_cast<bf16x8_t*>(dgu + t * 2 * I + c), og);
    st_stream<NT>(reinterpret_cast<bf16x8_t*>(dgu + t * 2 * I + I + c), ou);
  }
}

// =========================================================================== GELU (erf), bf16
// torch.nn.GELU() inside UniGen.mm_projector (reference models/unigen.py:119-128) under bf16 autocast
__global__ __launch_bounds__(256) void gelu_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                   bf16_t* __restrict__ out, int64_t n8) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    const bf16x8_t v = reinterpret_cast<const bf16x8_t*>(x)[i];
    bf16x8_t g, o;
    if (dy) g = reinterpret_cast<const bf16x8_t*>(dy)[i];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float f = bf2f((bf16_t)v[k]);
      const float cdf = 0.5f * (1.f + erff(f * 0.70710678118654752f));
      if (!dy) o[k] = (short)f2bf(f * cdf);
      else o[k] = (short)f2bf(bf2f((bf16_t)g[k]) * (cdf + f * 0.3989422804014327f * __expf(-0.5f * f * f)));
    }
    reinterpret_cast<bf16x8_t*>(out)[i] = o;
  }
}

// =========================================================================== embedding
__global__ __launch_bounds__(256) void embed_fwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ W,
                                                        float* __restrict__ out, int64_t tokens, int H, int64_t V,
                                                        int* __restrict__ err) {
  const int per_row = H >> 2;
  const int64_t total = tokens * per_row;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = idx / per_row; const int c = (int)(idx % per_row);
    int64_t id = ids[t];
    if (id < 0 || id >= V) { if (err) atomicOr(err, 1); id = 0; }
    reinterpret_cast<float4*>(out + t * H)[c] = reinterpret_cast<const float4*>(W + id * H)[c];
  }
}

__global__ __launch_bounds__(256) void embed_bwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ dout,
                                                        float* __restrict__ dW, int64_t tokens, int H, int64_t V) {
  const int64_t total = tokens * H;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = idx / H; const int c = (int)(idx % H);
    const int64_t id = ids[t];
    if (id >= 0 && id < V) atomicAdd(dW + id * H + c, dout[idx]);
  }
}

// Deterministic form for the data-parallel exchange (unigen_hip/ddp.py): the (id, gradient row) pairs of EVERY rank, sorted by id
// with a stable sort (so equal ids keep rank order, then position order).  One workgroup per sorted position; only the first
// position of a run of equal ids works: it sums the run's rows in that fixed order and adds scale * sum to dW[id] -- one writer
// per table row, no atomics, the same bits on every rank.
__global__ __launch_bounds__(256) void embed_bwd_sorted_kernel(const int64_t* __restrict__ ids_sorted, const int64_t* __restrict__ order,
                                                               const float* __restrict__ rows, float* __restrict__ dW,
                                                               int64_t n, int H, int64_t V, float scale) {
  const int64_t p = blockIdx.x;
  const int64_t id = ids_sorted[p];
  if (id < 0 || id >= V) return;
  if (p > 0 && ids_sorted[p - 1] == id) return;
  const int per_row = H >> 2;
  for (int c = threadIdx.x; c < per_row; c += blockDim.x) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t j = p; j < n && ids_sorted[j] == id; ++j) {
      const float4 v = reinterpret_cast<const float4*>(rows + order[j] * H)[c];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    float4* d = reinterpret_cast<float4*>(dW + id * H) + c;
    float4 o = *d;
    o.x += scale * acc.x; o.y += scale * acc.y; o.z += scale * acc.z; o.w += scale * acc.w;
    *d = o;
  }
}

// =========================================================================== column sums (bias grad)
// out[c] += sum_r in[r, c]   (bf16 in, fp32 out); block = 64 columns x 4 row-lanes
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* __restrict__ in, int64_t ld, float* __restrict__ out,
                                                     int R, int C, int rows_per_block) {
  __shared__ float red[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(R, r0 + rows_per_block);
  float s = 0.f;
  if (c < C) for (int r = r0 + ty; r < r1; r += 4) s += bf2f(in[(int64_t)r * ld + c]);
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && c < C) atomicAdd(out + c, red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx]);
}

// =========================================================================== AdamW (flat, fused)
// One element of the update, shared by every form of the kernel (vector loops, scalar tails, the pipelined small-grid form) with
// floating-point contraction OFF: the forms are bit-identical to one another by construction -- hipcc otherwise fuses
// multiply-adds differently in differently shaped code (a 1-ulp second moment in a scalar tail against the vector loop, found by
// `test_adamw_small_grid_form_is_the_same_update`).  Operation order = torch's single-tensor AdamW: mul_, lerp_, mul_ + addcmul_,
// sqrt / bias_correction2_sqrt + eps, addcdiv_.
struct AdamConsts { float grad_scale, decay, w1, beta2, w2, bc2_sqrt, eps, step_size; };
__device__ __forceinline__ void adamw_element(float& p, float g, float& m, float& v, const AdamConsts& c) {
#pragma clang fp contract(off)
  const float gr = g * c.grad_scale;
  const float pk = p * c.decay;
  const float mk = m + (gr - m) * c.w1;
  const float vk = v * c.beta2 + (c.w2 * gr) * gr;
  const float denom = sqrtf(vk) / c.bc2_sqrt + c.eps;
  p = pk - c.step_size * (mk / denom);
  m = mk; v = vk;
}
__device__ __forceinline__ AdamConsts adam_consts(float lr, float beta1, float beta2, float eps, float wd, float bc1, float bc2_sqrt, float grad_scale) {
#pragma clang fp contract(off)
  return AdamConsts{grad_scale, 1.f - lr * wd, 1.f - beta1, beta2, 1.f - beta2, bc2_sqrt, eps, lr / bc1};
}
// Same update order as torch.optim.AdamW's single-tensor path (reference optimizer,
// training/train.py:324-330): decay, lerp first moment, second moment, bias-corrected step.
// The overlapped form of the update (side stream, beside the tokenizer's convolutions): TWO elements per lane and iteration.
// The four-element kernel needs 50-53 registers; the eight-wave convolution workgroups hold 2 x 232 of a SIMD's 512, so one more
// wave fits beside them only up to 48 -- above that the update's workgroups wait for whole CUs and then hold them (a grid-stride
// workgroup lives for the whole update), and the convolutions run on what is left.  Same arithmetic, element by element.
template <int NT>
__global__ __launch_bounds__(256) void adamw_lean_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                         float* __restrict__ v, bf16_t* __restrict__ p_bf16, int64_t n, float lr,
                                                         float beta1, float beta2, float eps, float wd, float bc1, float bc2_sqrt,
                                                         float grad_scale) {
  const int64_t n2 = n >> 1;
  const AdamConsts c = adam_consts(lr, beta1, beta2, eps, wd, bc1, bc2_sqrt, grad_scale);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n2; i += (int64_t)gridDim.x * blockDim.x) {
    float2 pp = ld_stream<NT>(reinterpret_cast<const float2*>(p) + i);
    const float2 gg = ld_stream<NT>(reinterpret_cast<const float2*>(g) + i);
    float2 mm = ld_stream<NT>(reinterpret_cast<const float2*>(m) + i);
    float2 vv = ld_stream<NT>(reinterpret_cast<const float2*>(v) + i);
    float* pa = &pp.x; const float* ga = &gg.x; float* ma = &mm.x; float* va = &vv.x;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      adamw_element(pa[k], ga[k], ma[k], va[k], c);
    }
    {
    st_stream<NT>(reinterpret_cast<float2*>(p) + i, pp);
    st_stream<NT>(reinterpret_cast<float2*>(m) + i, mm);
    st_stream<NT>(reinterpret_cast<float2*>(v) + i, vv);
    if (p_bf16) reinterpret_cast<uint32_t*>(p_bf16)[i] = pack_bf2(pp.x, pp.y);
    }
  }
  const int64_t ti = (n2 << 1) + blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (ti < n) {
    float pk = p[ti], mk = m[ti], vk = v[ti];
    adamw_element(pk, g[ti], mk, vk, c);
    p[ti] = pk; m[ti] = mk; v[ti] = vk;
    if (p_bf16) p_bf16[ti] = f2bf(pk);
  }
}

// The lean kernel as a three-stage software pipeline (round 4): the loads of element pairs k+1 and k+2 are in flight while pair k is
// computed and stored.  One wave per SIMD is all that fits beside the convolutions, and with loads -> wait -> ~80 VALU operations ->
// stores in sequence that wave kept ONE request batch in flight: 11.2 ms alone against 7.05 for its memory streams without the
// arithmetic and 2.5 for the arithmetic without the streams (probe switch UG_ADAMW_ABLATE of tools/probes/probe_switches.patch), 22.1 ms beside the tokenizer.
// The loads are inline assembly with hand-counted waits: hipcc's wait-count pass treats loads and stores in flight together as
// unordered on gfx9 (one counter) and drains vmcnt(0) before the first use of any load issued ahead of a store.  vmcnt(8) is safe
// whatever the stores do: loads return in order among themselves, so at most eight operations outstanding means at most the eight
// younger loads (stages k+1, k+2) outstanding.  Stages are separate variables and the loop is unrolled by three: no register of
// a load in flight is ever copied.  Same arithmetic, element by element; 32-bit indices (n < 2^29), <= 48 registers.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
struct AdamStage { f32x2_t p, g, m, v; };
template <int NT>
__global__ __launch_bounds__(256) void adamw_lean2_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                          float* __restrict__ v, bf16_t* __restrict__ p_bf16, uint32_t n, float lr,
                                                          float beta1, float beta2, float eps, float wd, float bc1, float bc2_sqrt,
                                                          float grad_scale) {
  const uint32_t n2 = n >> 1;
  const AdamConsts c = adam_consts(lr, beta1, beta2, eps, wd, bc1, bc2_sqrt, grad_scale);
  const uint32_t stride = gridDim.x * blockDim.x;
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t trips = (n2 + stride - 1) / stride;             // uniform over the grid: waits and loads are never behind a divergent branch
  auto load = [&](AdamStage& s, uint32_t idx) {                   // clamped: lanes past the end re-read the last pair and drop it
    const uint32_t off = min(idx, n2 - 1) * 8u;
    asm volatile("global_load_dwordx2 %0, %1, %2 nt" : "=&v"(s.p) : "v"(off), "s"(p) : "memory");
    asm volatile("global_load_dwordx2 %0, %1, %2 nt" : "=&v"(s.g) : "v"(off), "s"(g) : "memory");
    asm volatile("global_load_dwordx2 %0, %1, %2 nt" : "=&v"(s.m) : "v"(off), "s"(m) : "memory");
    asm volatile("global_load_dwordx2 %0, %1, %2 nt" : "=&v"(s.v) : "v"(off), "s"(v) : "memory");
  };
  auto update = [&](AdamStage& s, uint32_t idx) {
    asm volatile("s_waitcnt vmcnt(8)" : "+v"(s.p), "+v"(s.g), "+v"(s.m), "+v"(s.v) : : "memory");   // ties the uses below to the wait
    if (idx < n2) {
      float2 pp, mm, vv;
      float* pa = &pp.x; float* ma = &mm.x; float* va = &vv.x;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        pa[k] = s.p[k]; ma[k] = s.m[k]; va[k] = s.v[k];
        adamw_element(pa[k], s.g[k], ma[k], va[k], c);
      }
      st_stream<NT>(reinterpret_cast<float2*>(p) + idx, pp);
      st_stream<NT>(reinterpret_cast<float2*>(m) + idx, mm);
      st_stream<NT>(reinterpret_cast<float2*>(v) + idx, vv);
      if (p_bf16) reinterpret_cast<uint32_t*>(p_bf16)[idx] = pack_bf2(pp.x, pp.y);
    }
  };
  if (n2 > 0) {
    AdamStage A, B, C;
    uint32_t i = tid;
    load(A, i);
    load(B, i + stride);
    for (uint32_t t = 0; t < trips; t += 3) {
      load(C, i + 2 * stride); update(A, i);
      load(A, i + 3 * stride); update(B, i + stride);
      load(B, i + 4 * stride); update(C, i + 2 * stride);
      i += 3 * stride;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  const uint32_t ti = (n2 << 1) + tid;
  if (ti < n) {
    float pk = p[ti], mk = m[ti], vk = v[ti];
    adamw_element(pk, g[ti], mk, vk, c);
    p[ti] = pk; m[ti] = mk; v[ti] = vk;
    if (p_bf16) p_bf16[ti] = f2bf(pk);
  }
}

// Also refreshes the bf16 compute copy of the weights in the same pass.
template <int NT>
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v,
                                                    bf16_t* __restrict__ p_bf16, int64_t n, float lr, float beta1,
                                                    float beta2, float eps, float wd, float bc1, float bc2_sqrt,
                                                    float grad_scale) {
  const int64_t n4 = n >> 2;
  const AdamConsts c = adam_consts(lr, beta1, beta2, eps, wd, bc1, bc2_sqrt, grad_scale);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 pp = ld_stream<NT>(reinterpret_cast<const float4*>(p) + i);
    float4 gg = ld_stream<NT>(reinterpret_cast<const float4*>(g) + i);
    float4 mm = ld_stream<NT>(reinterpret_cast<const float4*>(m) + i);
    float4 vv = ld_stream<NT>(reinterpret_cast<const float4*>(v) + i);
    float* pa = &pp.x; float* ga = &gg.x; float* ma = &mm.x; float* va = &vv.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      adamw_element(pa[k], ga[k], ma[k], va[k], c);
    }
    st_stream<NT>(reinterpret_cast<float4*>(p) + i, pp);
    st_stream<NT>(reinterpret_cast<float4*>(m) + i, mm);
    st_stream<NT>(reinterpret_cast<float4*>(v) + i, vv);
    if (p_bf16) { uint2 o; o.x = pack_bf2(pp.x, pp.y); o.y = pack_bf2(pp.z, pp.w); st_stream<NT>(reinterpret_cast<uint2*>(p_bf16) + i, o); }
  }
  // tail (n % 4)
  const int64_t tail0 = n4 << 2;
  const int64_t ti = tail0 + blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (ti < n) {
    float pk = p[ti], mk = m[ti], vk = v[ti];
    adamw_element(pk, g[ti], mk, vk, c);
    p[ti] = pk; m[ti] = mk; v[ti] = vk;
    if (p_bf16) p_bf16[ti] = f2bf(pk);
  }
}

__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, int64_t n) {
  const int64_t n4 = n >> 2;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(in)[i];
    uint2 o; o.x = pack_bf2(v.x, v.y); o.y = pack_bf2(v.z, v.w);
    reinterpret_cast<uint2*>(out)[i] = o;
  }
  const int64_t ti = (n4 << 2) + blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (ti < n) out[ti] = f2bf(in[ti]);
}

// =========================================================================== gradient exchange staging
// out = bf16(in * scale): the flat fp32 gradient bucket, pre-divided by the world size, packed for the bf16 all-reduce.
__global__ __launch_bounds__(256) void grad_pack_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, int64_t n, float scale) {
  const int64_t n8 = n >> 3;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 a = reinterpret_cast<const float4*>(in)[2 * i], b = reinterpret_cast<const float4*>(in)[2 * i + 1];
    uint4 o;
    o.x = pack_bf2(a.x * scale, a.y * scale); o.y = pack_bf2(a.z * scale, a.w * scale);
    o.z = pack_bf2(b.x * scale, b.y * scale); o.w = pack_bf2(b.z * scale, b.w * scale);
    reinterpret_cast<uint4*>(out)[i] = o;
  }
  const int64_t ti = (n8 << 3) + blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (ti < n) out[ti] = f2bf(in[ti] * scale);
}
// out[i] = bf16( (sum_{r < world} float(shards[r * stride + i])) * scale ): rank-ordered fp32 sum of the bf16 copies of ONE slice
// that an all-to-all collected from every rank (FlatGradSync reduce = "bf16_fp32acc": bf16 on the wire, fp32 arithmetic).
__global__ __launch_bounds__(256) void grad_sum_shards_kernel(const bf16_t* __restrict__ shards, int world, int64_t stride,
                                                              bf16_t* __restrict__ out, int64_t n, float scale) {
  const int64_t n8 = n >> 3;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < world; ++r) {
      const uint4 v = reinterpret_cast<const uint4*>(shards + r * stride)[i];
      acc[0] += __uint_as_float(v.x << 16); acc[1] += __uint_as_float(v.x & 0xffff0000u);
      acc[2] += __uint_as_float(v.y << 16); acc[3] += __uint_as_float(v.y & 0xffff0000u);
      acc[4] += __uint_as_float(v.z << 16); acc[5] += __uint_as_float(v.z & 0xffff0000u);
      acc[6] += __uint_as_float(v.w << 16); acc[7] += __uint_as_float(v.w & 0xffff0000u);
    }
    uint4 o;
    o.x = pack_bf2(acc[0] * scale, acc[1] * scale); o.y = pack_bf2(acc[2] * scale, acc[3] * scale);
    o.z = pack_bf2(acc[4] * scale, acc[5] * scale); o.w = pack_bf2(acc[6] * scale, acc[7] * scale);
    reinterpret_cast<uint4*>(out)[i] = o;
  }
  const int64_t ti = (n8 << 3) + blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (ti < n) {
    float a = 0.f;
    for (int r = 0; r < world; ++r) a += bf2f(shards[r * stride + ti]);
    out[ti] = f2bf(a * scale);
  }
}
// out = float(in): the reduced bf16 bucket back into the flat fp32 gradient buffer.
__global__ __launch_bounds__(256) void grad_unpack_kernel(const bf16_t* __restrict__ in, float* __restrict__ out, int64_t n) {
  const int64_t n8 = n >> 3;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    const uint4 v = reinterpret_cast<const uint4*>(in)[i];
    float4 a, b;
    a.x = __uint_as_float(v.x << 16); a.y = __uint_as_float(v.x & 0xffff0000u);
    a.z = __uint_as_float(v.y << 16); a.w = __uint_as_float(v.y & 0xffff0000u);
    b.x = __uint_as_float(v.z << 16); b.y = __uint_as_float(v.z & 0xffff0000u);
    b.z = __uint_as_float(v.w << 16); b.w = __uint_as_float(v.w & 0xffff0000u);
    reinterpret_cast<float4*>(out)[2 * i] = a; reinterpret_cast<float4*>(out)[2 * i + 1] = b;
  }
  const int64_t ti = (n8 << 3) + blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (ti < n) out[ti] = bf2f(in[ti]);
}

// =========================================================================== row gather / scatter
// out[i, :] = in[idx[i], :]  (bf16 rows, 16-byte chunks).  Used to pick the label positions that feed
// the lm_head GEMM and to route their gradient back.
__global__ __launch_bounds__(256) void gather_rows_kernel(const bf16_t* __restrict__ in, int64_t ld_in,
                                                          const int64_t* __restrict__ idx, bf16_t* __restrict__ out,
                                                          int64_t ld_out, int64_t n, int C, int scatter) {
  const int per_row = C >> 3;
  const int64_t total = n * per_row;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / per_row; const int c = (int)(i % per_row) * 8;
    const int64_t src = scatter ? r : idx[r], dst = scatter ? idx[r] : r;
    *reinterpret_cast<bf16x8_t*>(out + dst * ld_out + c) = *reinterpret_cast<const bf16x8_t*>(in + src * ld_in + c);
  }
}

inline int grid_for(int64_t work_items, int block = 256, int cap = 256 * 8) {
  static const int cap_env = [] { const char* e = getenv("UNIGEN_EW_GRID_CAP"); return e ? atoi(e) : 0; }();
  if (cap_env > 0) cap = cap_env;
  int64_t g = (work_items + block - 1) / block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

}  // namespace

// ------------------------------------------------------------------------------------ C ABI
extern "C" int ug_rmsnorm_fwd(const float* x, const float* w, void* y, float* rstd, int64_t rows, int64_t cols,
                              float eps, int out_f32, hipStream_t st) {
  UG_REQUIRE(rows > 0 && cols > 0 && cols % 4 == 0, "ug_rmsnorm_fwd: cols=%ld must be a positive multiple of 4", (long)cols);
  UG_REQUIRE(ug_aligned16(x) && ug_aligned16(w) && ug_aligned16(y), "ug_rmsnorm_fwd: pointers must be 16B aligned");
  dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  // rows per wave of the register form (0: the two-pass kernel).  In the step (tools/probes/ab.sh env): 27.7 us two-pass, 21.0 at one
  // row per wave, 22.2 at two, 25.4 at four
  static const int reg = [] { const char* e = getenv("UNIGEN_RN_FWD_REG"); return e ? atoi(e) : 1; }();
  if (reg > 0 && cols <= 2048 && !out_f32) {
    dim3 g2((unsigned)((rows + 4 * reg - 1) / (4 * reg)));
    hipLaunchKernelGGL((rmsnorm_fwd_reg_kernel<false, 8>), g2, block, 0, st, x, w, y, rstd, (int)rows, (int)cols, eps, reg);
  } else if (out_f32) hipLaunchKernelGGL(rmsnorm_fwd_kernel<true>, grid, block, 0, st, x, w, y, rstd, (int)rows, (int)cols, eps);
  else hipLaunchKernelGGL(rmsnorm_fwd_kernel<false>, grid, block, 0, st, x, w, y, rstd, (int)rows, (int)cols, eps);
  UG_CHECK_LAUNCH("ug_rmsnorm_fwd");
  return UG_OK;
}

extern "C" int ug_rmsnorm_bwd(const void* dy, const float* x, const float* rstd, const float* w, float* dres,
                              float* dw, void* dres_bf16, int64_t rows, int64_t cols, hipStream_t st) {
  UG_REQUIRE(rows > 0 && cols > 0 && cols % 4 == 0 && cols <= 64 * 4 * RN_MAXV,
             "ug_rmsnorm_bwd: cols=%ld unsupported (multiple of 4, <= %d)", (long)cols, 64 * 4 * RN_MAXV);
  UG_REQUIRE(ug_aligned16(x) && ug_aligned16(w) && ug_aligned16(dres) && ((uintptr_t)dy & 7) == 0,
             "ug_rmsnorm_bwd: pointers must be aligned");
  static const int rpb_env = [] { const char* e = getenv("UNIGEN_RN_RPB"); return e ? atoi(e) : 0; }();
  // ONE round of workgroups: the kernel's 182 registers leave two 4-wave workgroups per CU (512 slots), and 16 rows per workgroup --
  // round 3's choice, 771 workgroups for 12 336 rows -- is a full round plus a half-empty one.  Rows per workgroup = rows / 480 rounded
  // up to a multiple of 4 (28 at the benchmark shape: 441 workgroups, 7 rows per wave): element-wise family 15.5 -> 14.7 ms per step
  // (in-step sweep of 16 / 24 / 25 / 26 / 28 / 32 / 40 / 48: 15.5 / 16.1 / 14.8 / 14.8 / 14.7 / 14.6-14.8 / 15.0 / 15.6).
  const int rpb = rpb_env > 0 ? rpb_env : (int)std::max<int64_t>(16, ((rows + 479) / 480 + 3) / 4 * 4);
  dim3 grid((unsigned)((rows + rpb - 1) / rpb)), block(256);
  UG_REQUIRE(((uintptr_t)dres_bf16 & 7) == 0, "ug_rmsnorm_bwd: dres_bf16 must be 8-byte aligned");
  // (a 6-slot instantiation for 1536 columns -- 148 instead of 182 registers, three waves per SIMD instead of two -- measured
  // SLOWER inside the step: 86 vs 78 us per launch, profiles/r03b vs r03a; the generic form stays)
#define UG_RNB(NTV) hipLaunchKernelGGL((rmsnorm_bwd_kernel<RN_MAXV, NTV>), grid, block, 0, st, (const bf16_t*)dy, x, rstd, w, dres, dw, \
                                      (bf16_t*)dres_bf16, (int)rows, (int)cols, rpb)
  switch ((ew_nt() >> 8) & 0xf) {          // third hex digit
    case 1: UG_RNB(1); break;
    case 2: UG_RNB(2); break;
    case 3: UG_RNB(3); break;
    default: UG_RNB(0);
  }
#undef UG_RNB
  UG_CHECK_LAUNCH("ug_rmsnorm_bwd");
  return UG_OK;
}

extern "C" int ug_rope(void* qkv, const float* cos_tab, const float* sin_tab, int64_t tokens, int64_t L, int64_t ldq,
                       int nheads, int head_dim, int backward, hipStream_t st) {
  UG_REQUIRE(tokens > 0 && L > 0 && tokens % L == 0, "ug_rope: tokens=%ld must be a multiple of L=%ld", (long)tokens, (long)L);
  UG_REQUIRE(head_dim % 8 == 0 && ldq % 4 == 0, "ug_rope: head_dim must be a multiple of 8");
  UG_REQUIRE(((uintptr_t)qkv & 7) == 0 && ug_aligned16(cos_tab) && ug_aligned16(sin_tab), "ug_rope: alignment");
  const int64_t total = tokens * nheads * (head_dim / 8);
  dim3 grid(grid_for(total)), block(256);
  if (backward) hipLaunchKernelGGL(rope_kernel<true>, grid, block, 0, st, (bf16_t*)qkv, cos_tab, sin_tab, tokens, (int)L, (int)ldq, nheads, head_dim);
  else hipLaunchKernelGGL(rope_kernel<false>, grid, block, 0, st, (bf16_t*)qkv, cos_tab, sin_tab, tokens, (int)L, (int)ldq, nheads, head_dim);
  UG_CHECK_LAUNCH("ug_rope");
  return UG_OK;
}

extern "C" int ug_swiglu_fwd(const void* gate_up, void* act, int64_t tokens, int64_t I, hipStream_t st) {
  UG_REQUIRE(tokens > 0 && I % 8 == 0, "ug_swiglu_fwd: I=%ld must be a multiple of 8", (long)I);
  UG_REQUIRE(ug_aligned16(gate_up) && ug_aligned16(act), "ug_swiglu_fwd: alignment");
  dim3 grid(grid_for(tokens * (I / 8))), block(256);
  switch (ew_nt() & 0xf) {
    case 1: hipLaunchKernelGGL(swiglu_fwd_kernel<1>, grid, block, 0, st, (const bf16_t*)gate_up, (bf16_t*)act, tokens, (int)I); break;
    case 2: hipLaunchKernelGGL(swiglu_fwd_kernel<2>, grid, block, 0, st, (const bf16_t*)gate_up, (bf16_t*)act, tokens, (int)I); break;
    case 3: hipLaunchKernelGGL(swiglu_fwd_kernel<3>, grid, block, 0, st, (const bf16_t*)gate_up, (bf16_t*)act, tokens, (int)I); break;
    default: hipLaunchKernelGGL(swiglu_fwd_kernel<0>, grid, block, 0, st, (const bf16_t*)gate_up, (bf16_t*)act, tokens, (int)I);
  }
  UG_CHECK_LAUNCH("ug_swiglu_fwd");
  return UG_OK;
}

extern "C" int ug_swiglu_bwd(const void* gate_up, const void* dact, void* dgate_up, int64_t tokens, int64_t I,
                             hipStream_t st) {
  UG_REQUIRE(tokens > 0 && I % 8 == 0, "ug_swiglu_bwd: I=%ld must be a multiple of 8", (long)I);
  UG_REQUIRE(ug_aligned16(gate_up) && ug_aligned16(dact) && ug_aligned16(dgate_up), "ug_swiglu_bwd: alignment");
  dim3 grid(grid_for(tokens * (I / 8))), block(256);
#define UG_SWB(NTV) hipLaunchKernelGGL(swiglu_bwd_kernel<NTV>, grid, block, 0, st, (const bf16_t*)gate_up, (const bf16_t*)dact, (bf16_t*)dgate_up, tokens, (int)I)
  switch ((ew_nt() >> 4) & 0xf) {          // second hex digit: the backward
    case 1: UG_SWB(1); break;
    case 2: UG_SWB(2); break;
    case 3: UG_SWB(3); break;
    default: UG_SWB(0);
  }
#undef UG_SWB
  UG_CHECK_LAUNCH("ug_swiglu_bwd");
  return UG_OK;
}

extern "C" int ug_gelu(const void* x, const void* dy_or_null, void* out, int64_t n, hipStream_t st) {
  UG_REQUIRE(n > 0 && n % 8 == 0 && ug_aligned16(x) && ug_aligned16(out), "ug_gelu: n must be a multiple of 8, 16B-aligned buffers");
  dim3 grid(grid_for(n / 8)), block(256);
  hipLaunchKernelGGL(gelu_kernel, grid, block, 0, st, (const bf16_t*)x, (const bf16_t*)dy_or_null, (bf16_t*)out, n / 8);
  UG_CHECK_LAUNCH("ug_gelu");
  return UG_OK;
}

extern "C" int ug_embed_fwd(const int64_t* ids, const float* W, float* out, int64_t tokens, int64_t H, int64_t V,
                            int* err_flag, hipStream_t st) {
  UG_REQUIRE(tokens > 0 && H % 4 == 0, "ug_embed_fwd: H must be a multiple of 4");
  dim3 grid(grid_for(tokens * (H / 4))), block(256);
  hipLaunchKernelGGL(embed_fwd_kernel, grid, block, 0, st, ids, W, out, tokens, (int)H, V, err_flag);
  UG_CHECK_LAUNCH("ug_embed_fwd");
  return UG_OK;
}

extern "C" int ug_embed_bwd(const int64_t* ids, const float* dout, float* dW, int64_t tokens, int64_t H, int64_t V,
                            hipStream_t st) {
  UG_REQUIRE(tokens > 0 && H > 0, "ug_embed_bwd: empty");
  dim3 grid(grid_for(tokens * H)), block(256);
  hipLaunchKernelGGL(embed_bwd_kernel, grid, block, 0, st, ids, dout, dW, tokens, (int)H, V);
  UG_CHECK_LAUNCH("ug_embed_bwd");
  return UG_OK;
}

extern "C" int ug_embed_bwd_sorted(const int64_t* ids_sorted, const int64_t* order, const float* rows, float* dW, int64_t n,
                                   int64_t H, int64_t V, float scale, hipStream_t st) {
  UG_REQUIRE(ids_sorted && order && rows && dW && n > 0 && H > 0 && H % 4 == 0 && n < (1ll << 31) && ug_aligned16(rows) && ug_aligned16(dW),
             "ug_embed_bwd_sorted: need sorted ids, their permutation, 16B-aligned rows / table and H %% 4 == 0");
  hipLaunchKernelGGL(embed_bwd_sorted_kernel, dim3((unsigned)n), dim3(256), 0, st, ids_sorted, order, rows, dW, n, (int)H, V, scale);
  UG_CHECK_LAUNCH("ug_embed_bwd_sorted");
  return UG_OK;
}

extern "C" int ug_colsum_bf16(const void* in, int64_t ld, float* out, int64_t R, int64_t C, hipStream_t st) {
  UG_REQUIRE(R > 0 && C > 0, "ug_colsum_bf16: empty");
  const int rpb = 256;
  dim3 grid((unsigned)((C + 63) / 64), (unsigned)((R + rpb - 1) / rpb)), block(256);
  hipLaunchKernelGGL(colsum_kernel, grid, block, 0, st, (const bf16_t*)in, ld, out, (int)R, (int)C, rpb);
  UG_CHECK_LAUNCH("ug_colsum_bf16");
  return UG_OK;
}

extern "C" int ug_adamw_flat(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, float lr,
                             float beta1, float beta2, float eps, float weight_decay, int64_t step, float grad_scale,
                             int max_blocks, hipStream_t st) {
  UG_REQUIRE(n > 0 && step >= 1, "ug_adamw_flat: n=%ld step=%ld", (long)n, (long)step);
  UG_REQUIRE(ug_aligned16(p) && ug_aligned16(g) && ug_aligned16(m) && ug_aligned16(v) && ((uintptr_t)p_bf16 & 7) == 0,
             "ug_adamw_flat: buffers must be 16B aligned");
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  // max_blocks > 0: a deliberately small grid (one 4-wave workgroup per CU; the lean kernel above) for an update that runs on a
  // side stream beside MFMA-bound kernels -- it leaves the register file and LDS to them and lives off spare HBM bandwidth
  dim3 grid(max_blocks > 0 ? grid_for(n / 4 + 1, 256, max_blocks) : grid_for(n / 4 + 1)), block(256);
  static const int lean = [] { const char* e = getenv("UNIGEN_ADAMW_LEAN"); return e ? atoi(e) : 1; }();
  // round 4: the pipelined small-grid kernel is also the fastest form ALONE (tools/probes/adamw_alone.py, 400 M elements: 6.14 TB/s on 256
  // workgroups against 5.35 for the whole-chip four-element kernel), so the default grid (max_blocks = 0) takes it too
  static const int piped_default = [] { const char* e = getenv("UNIGEN_ADAMW_PIPED"); return e ? atoi(e) : 1; }();
  if (max_blocks == 0 && lean && piped_default && n < (1LL << 29)) max_blocks = 256;
  if (max_blocks > 0 && lean) {
    const int wgs = lean > 1 ? lean : max_blocks;
    dim3 lgrid(grid_for(n / 2 + 1, 256, wgs));
    static const int piped = [] { const char* e = getenv("UNIGEN_ADAMW_PIPED"); return e ? atoi(e) : 1; }();
    if (piped && n < (1LL << 29))
      hipLaunchKernelGGL(adamw_lean2_kernel<3>, lgrid, block, 0, st, p, g, m, v, (bf16_t*)p_bf16, (uint32_t)n, lr, beta1, beta2, eps, weight_decay,
                         (float)bc1, (float)sqrt(bc2), grad_scale);
    else if ((ew_nt() >> 12) & 0xf)
      hipLaunchKernelGGL(adamw_lean_kernel<3>, lgrid, block, 0, st, p, g, m, v, (bf16_t*)p_bf16, n, lr, beta1, beta2, eps, weight_decay,
                         (float)bc1, (float)sqrt(bc2), grad_scale);
    else
      hipLaunchKernelGGL(adamw_lean_kernel<0>, lgrid, block, 0, st, p, g, m, v, (bf16_t*)p_bf16, n, lr, beta1, beta2, eps, weight_decay,
                         (float)bc1, (float)sqrt(bc2), grad_scale);
    UG_CHECK_LAUNCH("ug_adamw_flat(lean)");
    return UG_OK;
  }
#define UG_ADW(NTV) hipLaunchKernelGGL(adamw_kernel<NTV>, grid, block, 0, st, p, g, m, v, (bf16_t*)p_bf16, n, lr, beta1, beta2, eps, \
                     weight_decay, (float)bc1, (float)sqrt(bc2), grad_scale)
  switch ((ew_nt() >> 12) & 0xf) {         // fourth hex digit
    case 1: UG_ADW(1); break;
    case 2: UG_ADW(2); break;
    case 3: UG_ADW(3); break;
    default: UG_ADW(0);
  }
#undef UG_ADW
  UG_CHECK_LAUNCH("ug_adamw_flat");
  return UG_OK;
}

extern "C" int ug_cast_f32_bf16(const float* in, void* out, int64_t n, hipStream_t st) {
  UG_REQUIRE(n > 0 && ug_aligned16(in) && ((uintptr_t)out & 7) == 0, "ug_cast_f32_bf16: bad args");
  dim3 grid(grid_for(n / 4 + 1)), block(256);
  hipLaunchKernelGGL(cast_f32_bf16_kernel, grid, block, 0, st, in, (bf16_t*)out, n);
  UG_CHECK_LAUNCH("ug_cast_f32_bf16");
  return UG_OK;
}

extern "C" int ug_grad_pack_bf16(const float* in, void* out, int64_t n, float scale, hipStream_t st) {
  UG_REQUIRE(n > 0 && ug_aligned16(in) && ug_aligned16(out), "ug_grad_pack_bf16: n > 0 and 16-byte aligned buffers required");
  dim3 grid(grid_for(n / 8 + 1)), block(256);
  hipLaunchKernelGGL(grad_pack_kernel, grid, block, 0, st, in, (bf16_t*)out, n, scale);
  UG_CHECK_LAUNCH("ug_grad_pack_bf16");
  return UG_OK;
}

// ranges[2 r] = first element, ranges[2 r + 1] = element count (both multiples of 4) of the r-th span of `buf` to clear
__global__ __launch_bounds__(256) void zero_ranges_kernel(float* __restrict__ buf, const int64_t* __restrict__ ranges) {
  const int64_t lo = ranges[2 * blockIdx.x], n4 = ranges[2 * blockIdx.x + 1] >> 2;
  float4* dst = reinterpret_cast<float4*>(buf + lo);
  for (int64_t i = blockIdx.y * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.y * blockDim.x)
    dst[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

extern "C" int ug_zero_ranges_f32(float* buf, const int64_t* ranges, int64_t n_ranges, int64_t max_len, hipStream_t st) {
  UG_REQUIRE(n_ranges > 0 && n_ranges < 65536 && max_len > 0 && ug_aligned16(buf) && ranges,
             "ug_zero_ranges_f32: 1..65535 ranges, a 16-byte aligned buffer and a device range table required");
  const unsigned per = (unsigned)std::min<int64_t>(64, (max_len / 4 + 255) / 256);
  hipLaunchKernelGGL(zero_ranges_kernel, dim3((unsigned)n_ranges, per), dim3(256), 0, st, buf, ranges);
  UG_CHECK_LAUNCH("ug_zero_ranges_f32");
  return UG_OK;
}

extern "C" int ug_grad_unpack_bf16(const void* in, float* out, int64_t n, hipStream_t st) {
  UG_REQUIRE(n > 0 && ug_aligned16(in) && ug_aligned16(out), "ug_grad_unpack_bf16: n > 0 and 16-byte aligned buffers required");
  dim3 grid(grid_for(n / 8 + 1)), block(256);
  hipLaunchKernelGGL(grad_unpack_kernel, grid, block, 0, st, (const bf16_t*)in, out, n);
  UG_CHECK_LAUNCH("ug_grad_unpack_bf16");
  return UG_OK;
}

extern "C" int ug_grad_sum_shards_bf16(const void* shards, int world, int64_t stride, void* out, int64_t n, float scale, hipStream_t st) {
  UG_REQUIRE(n > 0 && world >= 1 && world <= 1024 && stride >= n && stride % 8 == 0 && ug_aligned16(shards) && ug_aligned16(out),
             "ug_grad_sum_shards_bf16: n > 0, 1..1024 shards, stride >= n and a multiple of 8, 16-byte aligned buffers required");
  dim3 grid(grid_for(n / 8 + 1)), block(256);
  hipLaunchKernelGGL(grad_sum_shards_kernel, grid, block, 0, st, (const bf16_t*)shards, world, stride, (bf16_t*)out, n, scale);
  UG_CHECK_LAUNCH("ug_grad_sum_shards_bf16");
  return UG_OK;
}

extern "C" int ug_gather_rows_bf16(const void* in, int64_t ld_in, const int64_t* idx, void* out, int64_t ld_out,
                                   int64_t n, int64_t C, int scatter, hipStream_t st) {
  UG_REQUIRE(n > 0 && C % 8 == 0 && ld_in % 8 == 0 && ld_out % 8 == 0, "ug_gather_rows_bf16: C and strides must be multiples of 8");
  UG_REQUIRE(ug_aligned16(in) && ug_aligned16(out) && idx, "ug_gather_rows_bf16: alignment");
  dim3 grid(grid_for(n * (C / 8))), block(256);
  hipLaunchKernelGGL(gather_rows_kernel, grid, block, 0, st, (const bf16_t*)in, ld_in, idx, (bf16_t*)out, ld_out, n, (int)C, scatter);
  UG_CHECK_LAUNCH("ug_gather_rows_bf16");
  return UG_OK;
}
