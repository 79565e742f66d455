/* unigen_hip.h -- C ABI of libunigen_hip.so (gfx950 / MI355X).
 *
 * This is the drop-in boundary underneath the reference's Python model API:
 * the reference itself has no native code, every entry point below replaces a third-party kernel
 * the reference reaches through torch / transformers.  The "replaces" note on each entry cites the
 * reference call site (paths relative to the reference repo) whose arithmetic it implements.
 *
 * Conventions
 *   - plain C types only: device pointers, int64 sizes / strides (in ELEMENTS), float scalars,
 *     a hipStream_t.  No torch types.  bf16 tensors are passed as void* (raw 16-bit storage).
 *   - every function returns 0 on success or a negative error class (UG_ERR_*); the message is
 *     available from ug_last_error() (thread-local).  Nothing throws or aborts across the boundary.
 *   - every tensor is owned by the caller; the library never frees caller memory, allocates only one lazily
 *     created device scratch (k-sliced GEMM partials), never synchronises the device, and launches only on the
 *     stream it is given (so every entry is
 *     legal inside a hipGraph capture).
 *   - shape / alignment requirements are checked at entry and fail loudly (UG_ERR_ARG).
 */
#ifndef UNIGEN_HIP_H
#define UNIGEN_HIP_H

#include <stdint.h>
#include <hip/hip_runtime_api.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UG_ABI_VERSION 7

/* ---- library ---------------------------------------------------------------------------- */
const char* ug_last_error(void);
int ug_abi_version(void);

/* ---- handle ----------------------------------------------------------------------------------
 * The library's only allocation: an opaque per-stream workspace (192 MiB of device scratch for k-sliced GEMM partials).
 * ug_create may allocate and must be called outside stream capture; no op entry point allocates, frees, clears or
 * synchronises, so every op is legal inside a hipGraph capture.  One handle per concurrently used stream (SURVEY 8b:
 * "allocates only an opaque workspace/handle created by ug_create() / released by ug_destroy()"). */
typedef struct ug_handle ug_handle;
int ug_create(ug_handle** out);
int ug_destroy(ug_handle* h);

/* ---- dense contraction ------------------------------------------------------------------- */
/* C[M,N] = opA . opB^T, bf16 operands, fp32 accumulate.  Each operand is given either row-major
 * (k contiguous: A[M][K] lda, B[N][K] ldb) or k-major (A[K][M] lda, B[K][N] ldb; a_kmajor/b_kmajor = 1),
 * so forward (0,0), dgrad (0,1: B = W[N_out][K_in]) and wgrad (1,1: contraction over tokens) need no
 * transposed copies.  K is the true contraction length (no padding; see gemm_bf16.hip for the K%8 rule).
 * epilogue: 0 = bf16 out (+ optional bf16 bias[N])
 *           1 = fp32 out, C = (beta ? C : 0) + alpha * acc   (alpha read from device if non-null)
 *           2 = fp32 residual: C = resid + bf16round(acc)
 * replaces: nn.Linear inside transformers Qwen2Attention / Qwen2MLP (q,k,v,o,gate,up,down_proj) and
 *           UniGen's tied lm_head, models/unigen.py:274-287; their autograd dgrad / wgrad. */
#define UG_EPI_BF16 0
#define UG_EPI_F32 1
#define UG_EPI_RESID 2
int ug_gemm_bf16(const ug_handle* h, const void* A, int64_t lda, int a_kmajor, const void* B, int64_t ldb, int b_kmajor,
                 void* C, int64_t ldc, int64_t M, int64_t N, int64_t K, int epilogue, const void* bias,
                 const float* resid, int64_t ldr, int beta, const float* alpha_dev, int policy, hipStream_t stream);
/* h: the calling stream's handle (ug_create) or null.  The handle owns the fp32 scratch of the k-sliced launch forms
 * (partial last round of 256x256 tiles, small weight gradients, lm-head dgrad); without one those forms are not chosen.
 * One handle must not be used by two streams at once; any number of handles may run concurrently.
 * policy: -1 = automatic kernel selection (the product path); >= 0 pins a kernel for A/B benchmarks and tests:
 * 0 = 128x128 two LDS stages, 2 = 128x128 one LDS stage, 3 = staggered 256x256, 6 = 256x256 with the k-sliced tail forced,
 * 8 = k-sliced small outputs, 10 = 320x256 tiles (A row-major, N % 256 == 0, K % 32 == 0, bf16 / residual epilogue; other
 * launches ignore it), 44 ... 52 except 48 = the same kernel at a tile height of 16 * (policy - 32) = 192 ... 320 rows; UG_GEMM_POLICY_AUTO_BITS = automatic selection when only modifier bits are wanted;
 * | UG_GEMM_NARROW_EPILOGUE = element-wise instead of LDS-transposed 16-byte stores in the 256x256 kernel. */
#define UG_GEMM_NARROW_EPILOGUE 0x100
#define UG_GEMM_ONE_BARRIER 0x200     /* force the 256x256 kernel's one-barrier-per-k-tile main loop (default: weight gradients, both operands k-major) */
#define UG_GEMM_TWO_BARRIERS 0x400    /* force its two-barrier (L | M phase) main loop (default: forward and dgrad) */
#define UG_GEMM_POLICY_AUTO_BITS 0xff

int ug_cast_f32_bf16(const float* in, void* out, int64_t n, hipStream_t stream);

/* The down projection's dgrad with the SwiGLU backward in its epilogue (autograd of Qwen2MLP.forward, modeling_qwen2.py:46-48, under bf16
 * autocast; reference call site models/unigen.py:274-285 through loss.backward(), training/train.py:775): dgu[M, 2I] = d(gate | up) from
 * dy[M, K] (gradient of the down projection's output), W_down stored [K, I] (row stride ldw) and the forward's gu[M, 2I] -- the values of
 * ug_gemm_bf16 (B k-major) followed by ug_swiglu_bwd, bit for bit; d(act) never reaches HBM.  Needs I % 256 == 0, K % 32 == 0 and 16-byte
 * aligned rows; other shapes are refused (UG_ERR_ARG) and run as those two launches.  ABI v5. */
int ug_gemm_bf16_swiglu_bwd(const ug_handle* h, const void* dy, int64_t ld_dy, const void* w_down, int64_t ldw, const void* gu,
                            int64_t ld_gu, void* dgu, int64_t ld_dgu, int64_t M, int64_t I, int64_t K, hipStream_t stream);
/* Tile height of the three fused-epilogue launches above (128 ... 320 rows in steps of 16; 0 = automatic, the default): tests and
 * A/B runs force every instantiated height this way; a height the called entry point does not instantiate is refused (UG_ERR_ARG).
 * Process-wide.  ABI v6. */
int ug_gemm_set_fused_tile_height(int rows);
/* The fused q/k/v projection of a decoder layer (transformers modeling_qwen2.py:200-215 q_proj / k_proj / v_proj + apply_rotary_pos_emb
 * :131-135; reference call site models/unigen.py:274-285): qkv[M, N] = bf16(x[M, K] W[N, K]^T + bias) with rotate-half RoPE (tables
 * [L, head_dim / 2] fp32, row m at position m % L) applied in the GEMM's epilogue to the first rope_cols columns (the q and k heads) --
 * bit-identical to ug_gemm_bf16 followed by ug_rope.  Fused for head_dim 128, N / rope_cols / K multiples of 256 / 256 / 32; any other
 * shape runs as those two launches.  ABI v5. */
int ug_gemm_bf16_qkv_rope(const ug_handle* h, const void* x, int64_t ldx, const void* w, int64_t ldw, const void* bias, void* qkv,
                          int64_t ldq, int64_t M, int64_t N, int64_t K, const float* cos_tab, const float* sin_tab, int64_t L,
                          int64_t rope_cols, int head_dim, hipStream_t stream);
/* replaces: down_proj's input act_fn(gate_proj(x)) * up_proj(x) in Qwen2MLP.forward (modeling_qwen2.py:46-48) as ONE launch:
 * the gate_up projection of x [M, K] with the fused weight w_gate_up [2I, K] (gate rows, then up rows) whose epilogue also
 * writes act [M, I] = bf16(bf16(silu(gate)) * up).  gu [M, 2I] = [gate | up] is still written (the backward reads it).  A tile
 * of the 256x256 kernel takes 128 gate rows and the same hidden units' 128 up rows of the weight, so both values of a hidden
 * unit meet in one workgroup; values are bit-identical to ug_gemm_bf16 followed by ug_swiglu_fwd, which is also what this
 * entry runs for shapes the 256x256 kernel is not used for (I % 128 != 0 or fewer than 200 tiles). */
int ug_gemm_bf16_swiglu(const ug_handle* h, const void* x, int64_t ldx, const void* w_gate_up, int64_t ldw, void* gu,
                        int64_t ld_gu, void* act, int64_t ld_act, int64_t M, int64_t I, int64_t K, hipStream_t stream);

/* Grouped weight gradients (replaces: the autograd wgrad of the four nn.Linear of a Qwen2DecoderLayer, modeling_qwen2.py:
 * q/k/v_proj fused, o_proj, gate/up_proj fused, down_proj): dw[i][rows_i, cols_i] (= or +=, beta[i]) dy[i]^T x[i] over K[i]
 * tokens, both operands token-major (dy[i]: [K_i][ld_dy_i], x[i]: [K_i][ld_x_i], bf16), fp32 outputs with row stride ld_dw[i].
 * Tiles of one K cost the same, so ONE launch packs the problems into ceil(sum tiles / 256) rounds of the chip (3 for a
 * 1.5B layer at 12 336 tokens; 2 + 1 rounds + two k-sliced launches when issued one by one).  K is per problem (round 4): the
 * host appends a slice of the tied lm_head's weight gradient (models/unigen.py:287, K = the label rows) to every layer's
 * launch, where its short tiles fill the CUs the partial last round leaves idle.  Host arrays of n <= 8 entries. */
int ug_gemm_bf16_wgrad_group(int n, const void* const* dy, const int64_t* ld_dy, const void* const* x, const int64_t* ld_x,
                             float* const* dw, const int64_t* ld_dw, const int64_t* rows, const int64_t* cols,
                             const int* beta, const int64_t* K, hipStream_t stream);

/* ---- Qwen2 decoder-layer row ops ----------------------------------------------------------- */
/* replaces: transformers Qwen2RMSNorm.forward (modeling_qwen2.py:246-252) + the autocast bf16 cast
 * of the following Linear's input.  x fp32 [rows,cols]; y bf16 (or fp32 if out_f32); rstd optional. */
int ug_rmsnorm_fwd(const float* x, const float* w, void* y, float* rstd, int64_t rows, int64_t cols,
                   float eps, int out_f32, hipStream_t stream);
/* dres += d(rmsnorm)/dx . dy ; dw += sum_rows dy * xhat      (dy bf16); dres_bf16 (optional): bf16 copy of the
 * updated dres = the next GEMM's operand, written here instead of by a separate cast pass */
int ug_rmsnorm_bwd(const void* dy, const float* x, const float* rstd, const float* w, float* dres,
                   float* dw, void* dres_bf16, int64_t rows, int64_t cols, hipStream_t stream);
/* replaces: apply_rotary_pos_emb (modeling_qwen2.py:113-135); in place on `nheads` consecutive heads of
 * width head_dim starting at qkv (row stride ldq); position of row t is t % L; cos/sin [L, head_dim/2]. */
int ug_rope(void* qkv, const float* cos_tab, const float* sin_tab, int64_t tokens, int64_t L, int64_t ldq,
            int nheads, int head_dim, int backward, hipStream_t stream);
/* replaces: Qwen2MLP act_fn(gate) * up (modeling_qwen2.py:46-48); gate_up = [tokens, 2I] = [gate | up] */
int ug_swiglu_fwd(const void* gate_up, void* act, int64_t tokens, int64_t I, hipStream_t stream);
int ug_swiglu_bwd(const void* gate_up, const void* dact, void* dgate_up, int64_t tokens, int64_t I,
                  hipStream_t stream);
/* GELU(erf) of UniGen.mm_projector (models/unigen.py:119-128), bf16: out = gelu(x), or dgelu(x)*dy if dy given */
int ug_gelu(const void* x, const void* dy_or_null, void* out, int64_t n, hipStream_t stream);
/* replaces: llm.model.embed_tokens (models/unigen.py:257,370; fp32 gather) and its scatter-add grad */
int ug_embed_fwd(const int64_t* ids, const float* W, float* out, int64_t tokens, int64_t H, int64_t V,
                 int* err_flag, hipStream_t stream);
int ug_embed_bwd(const int64_t* ids, const float* dout, float* dW, int64_t tokens, int64_t H, int64_t V,
                 hipStream_t stream);
/* the same scatter-add, deterministic, for the (id, row) pairs gathered from every data-parallel rank: `ids_sorted` ascending
 * (stable sort: equal ids in rank, then position order; ids outside [0, V) are padding and skipped), `order[p]` = row of `rows`
 * that sorted position p came from; dW[id] += scale * (sum of the run's rows in that order), one writer per table row */
int ug_embed_bwd_sorted(const int64_t* ids_sorted, const int64_t* order, const float* rows, float* dW, int64_t n,
                        int64_t H, int64_t V, float scale, hipStream_t stream);
/* gather (scatter=0): out[i,:] = in[idx[i],:] ; scatter (1): out[idx[i],:] = in[i,:]   (bf16 rows)
 * replaces the logits[..., -(n+1):-1] / [:, :-1] position slicing of models/unigen.py:310-338 */
int ug_gather_rows_bf16(const void* in, int64_t ld_in, const int64_t* idx, void* out, int64_t ld_out,
                        int64_t n, int64_t C, int scatter, hipStream_t stream);
/* out[c] += sum_r in[r,c]  (bias gradients) */
int ug_colsum_bf16(const void* in, int64_t ld, float* out, int64_t R, int64_t C, hipStream_t stream);

/* ---- attention ------------------------------------------------------------------------------ */
/* Mask compression.  replaces the dense additive [B,1,L,L] masks of training/prompting_utils.py:975-1074
 * as consumed by SDPA: bits[B][L][nW] (nW = ceil(L/64); bit j of word w set <=> key w*64+j visible),
 * tileany[B][nW][nW].  err_flag bit 1 is raised if a value is neither 0 nor <= -1e9. */
#define UG_MASK_F32 0
#define UG_MASK_BF16 1
#define UG_MASK_I64 2
#define UG_MASK_BOOL 3
int ug_attn_mask_compress(const void* mask, int mask_dtype, int64_t stride_b, int64_t stride_row,
                          uint64_t* bits, uint8_t* tileany, int64_t B, int64_t L, int* err_flag,
                          hipStream_t stream);
int ug_attn_mask_causal(const uint8_t* key_valid /* [B,L] or null */, uint64_t* bits, uint8_t* tileany,
                        int64_t B, int64_t L, hipStream_t stream);
/* replaces: torch SDPA in Qwen2Attention.forward (modeling_qwen2.py:196-234), GQA H:HKV, head_dim 128.
 * q/k/v: row (b*L+t), head h at column h*128, row stride ldq.
 * o: [tokens, ldo] bf16, lse: [B][H][L] fp32. */
int ug_attn_fwd(const void* q, const void* k, const void* v, int64_t ldq, void* o,
                int64_t ldo, float* lse, const uint64_t* bits, const uint8_t* tileany, int64_t B, int64_t L,
                int64_t Lp, int H, int HKV, int head_dim, float scale, hipStream_t stream);
/* dq/dk/dv written with row stride ldg (heads laid out like q/k/v); delta: [B][H][L] workspace.
 * dkv_ws: optional fp32 [B*L][2*HKV*128] workspace, ZERO on entry and left zero: dK/dV are then accumulated per query
 * head (6x the workgroups, balanced under causal masks) with fp32 atomics; null -> per-kv-head kernel, no atomics. */
int ug_attn_bwd(const void* q, const void* k, const void* v, int64_t ldq,
                const void* o, const void* dout, int64_t ldo, const float* lse,
                float* delta, void* dq, void* dk, void* dv, int64_t ldg, const uint64_t* bits,
                const uint8_t* tileany, int64_t B, int64_t L, int64_t Lp, int H, int HKV, int head_dim,
                float scale, float* dkv_ws, const float* rope_cos, const float* rope_sin, float* dbias_qkv, hipStream_t stream);
/* rope_cos / rope_sin ([L][64] fp32, or null): dq and dk leave as the gradient w.r.t. the PRE-RoPE projections -- the transposed
 * rotation of apply_rotary_pos_emb (modeling_qwen2.py:131-135) applied where the gradients are stored, same arithmetic as
 * ug_rope(backward = 1) on the stored tensor.  dbias_qkv ([(H + 2 HKV) * 128] fp32, or null): += the column sums of the stored
 * dq | dk | dv (the bias gradient of the fused q / k / v projection, what ug_colsum_bf16 over the three tensors adds). */

/* ---- autoregressive decode (static KV cache, graph-capturable) -------------------------------- */
/* replaces: transformers DynamicCache.update + SDPA on one new token per row inside
 * UniGen.t2i_generate_ar (models/unigen.py:485-519).  Cache layout K,V [rows][HKV][Tmax][128] bf16.
 * Positions / lengths are read from DEVICE ints so a captured graph can be replayed for every step. */
int ug_kv_store(const void* qkv, int64_t ldq, int64_t k_col, int64_t v_col, void* cache_k, void* cache_v,
                int64_t rows, int64_t L, int HKV, int head_dim, int64_t Tmax, const int* pos_dev, int pos_host,
                hipStream_t stream);
int ug_rope_at(void* qkv, const float* cos_tab, const float* sin_tab, int64_t rows, int64_t ldq, int nheads,
               int head_dim, const int* pos_dev, int64_t max_pos, hipStream_t stream);
int ug_attn_decode(const void* q, int64_t ldq, const void* cache_k, const void* cache_v, const uint8_t* key_valid,
                   void* o, int64_t ldo, int64_t rows, int H, int HKV, int head_dim, int64_t Tmax,
                   const int* len_dev, float scale, hipStream_t stream);
/* weight-streaming GEMV for <= 32 activation rows: acc[r*acc_stride_r + n*acc_stride_n] += sum_k x[r][k] W[n][k]
 * (fp32 atomics into a zeroed accumulator; the decode-time form of every nn.Linear of the backbone,
 * models/unigen.py:496-502).  The decode path keeps row-major accumulators (acc_stride_n = 1). */
int ug_gemv_bf16(const void* x, int64_t ldx, int64_t R, const void* W, int64_t ldw, float* acc, int64_t acc_stride_r,
                 int64_t acc_stride_n, int64_t N, int64_t K, hipStream_t stream);
/* last launch of a decode step (Qwen2Model.norm after the last layer's down_proj, models/unigen.py:496-502): consumes the
 * row-major fp32 accumulator acc[r*ldacc + n] a decode GEMV filled and leaves it zeroed for the next step:
 *   x += bf16round(acc);  xn = bf16(rmsnorm(x) * w) */
int ug_decode_finish_resid_norm(float* acc, int64_t ldacc, float* x, const float* w, void* xn, int64_t rows, int64_t cols,
                                float eps, int* pos_inc, int* len_inc, hipStream_t stream);
/* pos_inc / len_inc (both or neither): the cache's device-side write position and visible length, incremented by one here -- every
 * kernel of the step that reads them has run by then (the `cache_position += 1` of the reference's loop, models/unigen.py:517). */
/* Five-launch decode layer (what UniGen.t2i_generate_ar's per-token forward, models/unigen.py:496-502, runs).  A decode
 * step is bound by the ~4 us floor of every launch, so all finishing work moves to the CONSUMER of each accumulator
 * and kernel boundaries are the only synchronisation:
 *   ug_decode_gemv_resid_norm  q/k/v and gate/up projections.  Operand = bf16(norm_w[k] * xnew[r][k]) with
 *        xnew = x_in + bf16round(pending) (the previous projection's residual add); writes xnew to x_out (!= x_in) and
 *        adds sum_k xnew^2 into ss_out[r].  RMSNorm's per-row rsqrt factor is applied by the accumulator's consumer.
 *   ug_attn_decode_fused       builds the new token's q/k/v from the raw qkv accumulator (rstd, bias, RoPE), appends k/v
 *        to the cache at *pos_dev, attends to cache keys [0, *pos_dev) + the new token.
 *   ug_decode_gemv             o projection (bf16 operand).
 *   ug_decode_gemv_swiglu      down projection.  Operand = bf16(bf16(silu(g)) * u), g, u = bf16(rstd[r] * gate/up acc).
 * Each GEMV also clears up to two fully-consumed accumulators (zero0/zero1, n floats each, multiples of 4) and one
 * 32-float statistics slot, so a captured step has no memset nodes. */
int ug_decode_gemv(const void* x, int64_t ldx, int64_t R, const void* W, int64_t ldw, float* acc, int64_t ldacc, int64_t N,
                   int64_t K, float* zero0, int64_t n0, float* zero1, int64_t n1, float* ss_zero, hipStream_t stream);
int ug_decode_gemv_resid_norm(const float* x_in, const float* pending, int64_t ld_pending, const float* norm_w, float* x_out,
                              float* ss_out, int64_t R, const void* W, int64_t ldw, float* acc, int64_t ldacc, int64_t N,
                              int64_t K, float* zero0, int64_t n0, float* zero1, int64_t n1, float* ss_zero,
                              hipStream_t stream);
int ug_decode_gemv_swiglu(const float* gate_up_acc, int64_t ld_gu, const float* ss_in, float eps, int64_t norm_cols, int64_t R,
                          const void* W, int64_t ldw, float* acc, int64_t ldacc, int64_t N, int64_t K, float* zero0, int64_t n0,
                          float* zero1, int64_t n1, float* ss_zero, hipStream_t stream);
int ug_attn_decode_fused(const float* acc_qkv, int64_t ldacc, const float* ss_in, float eps, int64_t norm_cols, const void* bias,
                         const float* cos_tab, const float* sin_tab, const int* pos_dev, void* cache_k, void* cache_v,
                         const uint8_t* key_valid, void* o, int64_t ldo, int64_t rows, int H, int HKV, int head_dim,
                         int64_t Tmax, int64_t max_pos, float scale, hipStream_t stream);
/* Single-writer decode projections (round 6; the same reference call sites: models/unigen.py:496-502 per-token forward, transformers
 * modeling_qwen2.py Qwen2DecoderLayer.forward / Qwen2MLP.forward / Qwen2Model.norm + lm_head).  Every output element has ONE writer -- a
 * workgroup owns a few weight rows for the whole contraction -- so these launches use no fp32 atomics and leave FINISHED values:
 *   ug_decode_sw_resid    h[r][n] += float(bf16(sum_k x[r][k] W[n][k]))  in place (o projection on the attention output; K == 1536)
 *   ug_decode_sw_gate_up  act = bf16(bf16(silu(gate)) * up), gate | up = Linear(RMSNorm(h))   (Qwen2MLP.forward; W = [2 I][H], gate rows first)
 *   ug_decode_sw_head     logits fp32 [R][N] = Linear(RMSNorm(h)) for N rows of the tied embedding; pos_inc / len_inc (both or
 *                         neither) are incremented by one (the step's last reader of them has run)
 * `h` is the fp32 residual stream [R][H]; RMSNorm is applied as the reference applies it: operand = bf16(norm_w * (h * rsqrt(mean(h^2)
 * + eps))).  The norm-fed entry points also sit behind a split-K producer of the form above: with `pend` (raw fp32 accumulator
 * [R][ld_pend]) the residual stream is h + float(bf16(pend)), written to x_out (a buffer other than h, or NULL) by the launch's first
 * eight workgroups; the accumulator is left as it is (a later launch clears it).
 * R <= 16; ug_decode_sw_supported() says whether a model's sizes fit this build (hidden 1536 = six 256-wide k-slabs, head_dim 128). */
int ug_decode_sw_supported(int64_t hidden, int64_t inter, int64_t q_dim, int head_dim);
/* The down projection's form (its 287 KB operand rules a whole-K workgroup out): split-K in k-blocks of seven 256-wide slabs -- a
 * workgroup owns 32 weight rows x one k-block, its seven partial tiles meet in LDS and join acc[r*ldacc + n] by ONE fp32 atomic per
 * element (K / 1792 atomics per output instead of K / 256).  Same accumulator / clears contract as ug_decode_gemv above. */
int ug_decode_sw_kblock(const void* x, int64_t ldx, int64_t R, const void* W, int64_t ldw, float* acc, int64_t ldacc, int64_t N, int64_t K,
                        float* zero0, int64_t n0, float* zero1, int64_t n1, float* ss_zero, hipStream_t stream);
int ug_decode_sw_resid(const void* x, int64_t ldx, int64_t R, const void* W, int64_t ldw, int64_t N, int64_t K, float* h,
                       hipStream_t stream);
int ug_decode_sw_gate_up(const float* h, const float* pend, int64_t ld_pend, float* x_out, const float* norm_w, float eps, int64_t R,
                         int64_t H, const void* W, int64_t ldw, int64_t I, void* act, int64_t ld_act, hipStream_t stream);
int ug_decode_sw_head(const float* h, const float* pend, int64_t ld_pend, float* x_out, const float* norm_w, float eps, int64_t R,
                      int64_t H, const void* W, int64_t ldw, int64_t N, float* logits, int64_t ld_logits, int* pos_inc, int* len_inc,
                      hipStream_t stream);
/* finish a split-K fp32 accumulation: mode 0: out_bf16 = bf16(acc + bias); mode 1: resid += bf16round(acc) */
int ug_skinny_finish(const float* acc, const void* bias, void* out_bf16, float* resid, int64_t M, int64_t N,
                     int mode, hipStream_t stream);

/* ---- device-side prompt / mask assembly --------------------------------- */
/* replaces the per-sample Python loops of UniversalPromptingQwen2.t2i_prompt (training/prompting_utils.py:59-111, prompt
 * dropout excluded) and the dense [B,1,L,L] masks of create_attention_mask_predict_next / _for_mmu (:975-1036): the
 * compressed mask is built straight from the token ids.  text_ids: all prompts back to back, text_offsets [B+1].
 * mode: 0 = predict_next with rm_pad_in_image (t2i), 1 = predict_next (lm), 2 = mmu (eoi of the first match in the batch).
 * meta_ws: int32 [4*B], flags_ws: bytes [B*L]. */
int ug_t2i_assemble(const int64_t* text_ids, const int64_t* text_offsets, const int64_t* conv_start, int64_t n_start,
                    const int64_t* conv_end, int64_t n_end, const int64_t* image_in, const int64_t* image_labels, int64_t B,
                    int64_t n_image, int64_t max_seq_len, int64_t pad_id, int64_t soi_id, int64_t eoi_id, int64_t ignore_id,
                    int64_t* input_ids, int64_t* labels, uint8_t* attn01, hipStream_t stream);
int ug_attn_mask_from_ids(const int64_t* ids, int64_t B, int64_t L, int64_t pad_id, int64_t soi_id, int64_t eoi_id, int mode,
                          int* meta_ws, uint8_t* flags_ws, uint64_t* bits, uint8_t* tileany, hipStream_t stream);

/* replaces: mask_or_random_replace_tokens (data/masking.py:13-94, the branch every shipped config takes: noise_type
 * 'mask', no contiguous region, predict_all_tokens off).  scores [B, n] fp32 are the caller's `torch.rand(B, n)`;
 * num_masked [B] fp32 = (n * mask_prob).round().clamp(min=1).  Position j is masked iff argsort(scores)[j] < num_masked
 * (ties broken by index).  input_ids = mask_id on masked positions else the token; labels = token on masked positions
 * else ignore_id.  Bit-equal to the reference for distinct scores. */
int ug_maskgit_train_mask(const int64_t* tokens, const float* scores, const float* num_masked, int64_t B, int64_t n,
                          int64_t mask_id, int64_t ignore_id, int64_t* input_ids, int64_t* labels, hipStream_t stream);

/* ---- MaskGIT parallel decoding step ------------------------------------------------------------ */
/* replaces, per round of UniGen.t2i_generate (models/unigen.py:404-451): the CFG mix of the code-book logits, softmax,
 * torch.multinomial, the gather of the drawn token's probability and models/sampling.py:41-46 mask_by_random_topk.
 * logits: bf16 [(cfg ? 2 : 1) * N * n][ld], row (b*n + i) = image b / position i, the N*n unconditional rows after the
 * conditional ones; V code-book columns.  u_sample / u_conf: uniforms in [0,1) [N*n] from the caller's generator
 * (token = inverse CDF in index order; Gumbel = -log(-log(u))).  cur_ids: code ids or mask_id [N*n].
 * mask_len_sched = floor(n * schedule(ratio)); temperature = the already-compounded Gumbel temperature.
 * Outputs [N*n]: sampled (known tokens kept), next_cur (mask_id where re-masked), next_ids (+ id_offset unless masked),
 * masking_out (optional bytes).  sel_ws: fp32 [N*n] scratch. */
int ug_maskgit_step(const void* logits, int64_t ld, int64_t V, int64_t N, int64_t n, int cfg, float guidance_scale,
                    const float* u_sample, const float* u_conf, const int64_t* cur_ids, int64_t mask_id, int64_t id_offset,
                    int64_t mask_len_sched, float temperature, int64_t* sampled, float* sel_ws, int64_t* next_cur,
                    int64_t* next_ids, uint8_t* masking_out, hipStream_t stream);

/* one autoregressive sampling step (models/unigen.py:503-519) on the raw fp32 lm-head accumulator acc [2*bsz][ldacc]
 * (conditional rows, then unconditional; rounded to bf16 like the head's output; cleared on exit): CFG mix, /temperature,
 * softmax + inverse-CDF draw on uniforms[step*bsz + b] (or argmax if greedy), step = *pos_dev - pos0.  Writes tok[b],
 * out_tokens[b*nsteps + step] and the next input x[b], x[bsz+b] = embed[token + id_offset] (fp32 rows of ld_embed). */
int ug_ar_sample(float* acc, int64_t ldacc, int64_t bsz, int64_t V, float guidance_scale, float temperature, int greedy,
                 const float* uniforms, const int* pos_dev, int64_t pos0, int64_t nsteps, const float* embed, int64_t ld_embed,
                 int64_t H, int64_t id_offset, int64_t* tok, int* out_tokens, float* x, hipStream_t stream);

/* ---- loss ------------------------------------------------------------------------------------ */
/* replaces: F.cross_entropy(ignore_index=-100) x3 in UniGen.forward (models/unigen.py:310-338) and
 * get_batch_logps (training/train_dpo.py:51-90).  logits bf16 [R, ld], ld % 8 == 0.
 * loss_and_count (optional): [0] = mean loss over valid rows, [1] = number of valid rows. */
int ug_ce_fwd(const void* logits, int64_t ld, int64_t R, int64_t V, const int64_t* labels,
              int64_t ignore_index, float* lse, float* loss_row, float* logp_label, float* loss_and_count,
              hipStream_t stream);
/* in place: logits <- (softmax - onehot) * (*gscale / count) (valid rows), 0 elsewhere incl. pad cols.
 * row_scale (optional, fp32 [R]): per-row factor instead of *gscale / count -- the backward of per-row label
 * log-probabilities (get_batch_logps, training/train_dpo.py:51-90: pass minus the upstream gradient of each row's logp) */
int ug_ce_bwd(void* logits_inout, int64_t ld, int64_t R, int64_t V, const int64_t* labels,
              int64_t ignore_index, const float* lse, const float* loss_and_count, const float* gscale,
              const float* row_scale, hipStream_t stream);

/* ---- optimizer ------------------------------------------------------------------------------- */
/* replaces: torch.optim.AdamW.step (training/train.py:324-330,780) over one flat fp32 segment; also
 * refreshes the bf16 compute copy (p_bf16 may be null).  grads are multiplied by grad_scale first. */
int ug_adamw_flat(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, float lr,
                  float beta1, float beta2, float eps, float weight_decay, int64_t step, float grad_scale,
                  int max_blocks /* 0 = fill the chip; > 0 = grid cap for an update overlapped with other kernels */,
                  hipStream_t stream);

/* ---- data-parallel gradient exchange staging ------------------------------------------------ */
/* replaces: the bucket copies + fp32 all-reduce of torch DistributedDataParallel's reducer behind
 * accelerator.backward (training/train.py:492,775).  A bucket of the flat fp32 gradient buffer is packed as
 * bf16(g * scale) (scale = 1 / world size: the mean is formed by the SUM all-reduce), exchanged by RCCL on the side
 * stream, and unpacked back in place.  Buffers 16-byte aligned; tails of any length. */
int ug_grad_pack_bf16(const float* in, void* out_bf16, int64_t n, float scale, hipStream_t stream);
/* Clear n_ranges spans of one fp32 buffer in ONE launch: ranges (device, int64) = {first element, element count} pairs, both
 * multiples of 4; max_len = the longest count (sizes the grid).  Start of a gradient pass: the small accumulating gradients
 * (norm weights, biases) that lie between the big "first write overwrites" matrices of the flat buffer -- 85 fill launches
 * per step otherwise (torch's `zero_grad(set_to_none=False)` semantics for those tensors, reference training/train.py:773). */
int ug_zero_ranges_f32(float* buf, const int64_t* ranges, int64_t n_ranges, int64_t max_len, hipStream_t stream);
int ug_grad_unpack_bf16(const void* in_bf16, float* out, int64_t n, hipStream_t stream);
/* bf16 on the wire, fp32 arithmetic: after an all-to-all of the packed bucket, `shards` holds every rank's bf16 copy of THIS
 * rank's slice (shard r at shards + r * stride elements, stride % 8 == 0).  out[i] = bf16(scale * sum_r float(shards[r][i])),
 * summed in fp32 in rank order (identical on every rank), rounded once; the slices are then all-gathered and unpacked. */
int ug_grad_sum_shards_bf16(const void* shards_bf16, int world, int64_t stride, void* out_bf16, int64_t n, float scale,
                            hipStream_t stream);

/* ---- data-parallel gradient exchange over RCCL ------------------------------------------------ */
/* replaces: the DistributedDataParallel reducer behind accelerator.prepare / accelerator.backward (training/train.py:492,775)
 * for hosts that do not bring torch.distributed (the shipped Python host keeps torch.distributed as its default transport and
 * switches to these entry points with UNIGEN_DDP_TRANSPORT=ug_comm).  One ug_comm per rank = one process per GPU.  RCCL is
 * resolved with dlopen at ug_comm_init time (the copy already mapped into the process, else UNIGEN_RCCL_LIB / librccl.so.1).
 *   ug_comm_unique_id       rank 0 creates the 128-byte rendezvous id; the host hands it to every rank (any side channel)
 *   ug_comm_init            ncclCommInitRank + the communicator's side stream, events and bf16 staging for buckets of up to
 *                           max_bucket_elems elements
 *   ug_comm_allreduce_bucket  MEAN over ranks of grad[0..n) in place.  Ordered after everything queued on `producer` at the
 *                           time of the call (an event is recorded there); pack / collective(s) / unpack run on the
 *                           communicator's side stream, so the producer stream continues with backward.  mode:
 *                             UG_COMM_FP32          ncclAllReduce(fp32, AVG): DDP's arithmetic, 4 bytes per element on the links
 *                             UG_COMM_BF16_FP32ACC  bf16 on the links, fp32 sum in rank order, one final bf16 rounding
 *                             UG_COMM_BF16          bf16(g / world) summed by ncclAllReduce in bf16 (least accurate)
 *                             UG_COMM_FP32_RSAG     fp32 as ncclReduceScatter(AVG) + ncclAllGather in place (+ a short all-reduce
 *                                                   for the tail that does not divide by the world size)
 *   ug_comm_allgather       recv[r * bytes_per_rank ...] = rank r's `send` bytes, on the communicator's side stream, ordered
 *                           after `producer` like a bucket (the per-token embedding-lookup gradient rows and their ids: the
 *                           dense head gradient of the tied table is exchanged right after the head's backward, the lookups'
 *                           few rows at the end, training/train.py:602-609 are the lookups)
 *   ug_comm_wait            `consumer` waits for every bucket issued so far (end of backward, before clipping / the optimizer)
 *   ug_comm_bytes_on_wire   payload handed to the collectives since init (reporting)
 * No call synchronises the host.  Buckets must be 16-byte aligned. */
typedef struct ug_comm ug_comm;
#define UG_COMM_ID_BYTES 128
#define UG_COMM_FP32 0
#define UG_COMM_BF16_FP32ACC 1
#define UG_COMM_BF16 2
#define UG_COMM_FP32_RSAG 3
int ug_comm_unique_id(void* id128);
int ug_comm_init(ug_comm** out, int world, int rank, const void* id128, int64_t max_bucket_elems);
int ug_comm_allreduce_bucket(ug_comm* comm, float* grad, int64_t n, int mode, hipStream_t producer);
int ug_comm_allgather(ug_comm* comm, const void* send, void* recv, int64_t bytes_per_rank, hipStream_t producer);
int ug_comm_wait(ug_comm* comm, hipStream_t consumer);
int ug_comm_destroy(ug_comm* comm);
int64_t ug_comm_bytes_on_wire(const ug_comm* comm);

/* ---- MAGVITv2 tokenizer (fp32, NHWC) ------------------------------------------------------- */
/* replaces: torch.nn.Conv2d in VQGANEncoder/Decoder, ResnetBlock, Downsample (asymmetric pad via
 * pad_top/pad_left = 0, stride 2), Upsample (nearest-2x folded into the load), magvitv2.py:90-178,
 * 319-408; common_modules.py:30-93,301-360.  w_packed: [k*k][Cin][cout_pad] fp32. */
int ug_conv2d_f32(const float* x, const float* w_packed, const float* bias, const float* residual, float* y,
                  int64_t B, int Hin, int Win, int Cin, int Cout, int cout_pad, int ksize, int stride,
                  int pad_top, int pad_left, int Hout, int Wout, int upsample2x, hipStream_t stream);
/* The same convolution at fp32 accuracy on the f16 matrix cores.  Each operand tensor is scaled by the power of two that
 * puts its largest magnitude into [2^14, 2^15), every scaled value split into two fp16 terms (a = a1 + a2, 23 of fp32's 24
 * significand bits) and a.b summed from the three partial products a1b1 + a1b2 + a2b1 in fp32 accumulators; the epilogue
 * undoes the scales.  Error vs fp64 equals a plain fp32 accumulation's (3.5e-7 relative at K = 1152; the reference's own GPU
 * path runs these convs with TF32 operands, 7.7e-4); 3/16 the MFMA cost of ug_conv2d_f32.
 * ug_conv_split_weights turns ug_conv2d_f32's packed weights into the split tile image once per weight version: w_split
 * holds 2 * taps * roundup(Cin, 32) * cout_pad 16-bit elements (a ragged last 32-channel slab is zero-filled) + 8 more
 * whose first four bytes record max|w| (computed here).  x_amax: device fp32, ANY upper bound of max|x| of the activation
 * tensor (ug_amax_f32 computes the exact one; a constant bound is as good -- only its binade is used); null = unscaled
 * (|x| < 65504 assumed, values saturate otherwise).  Needs Cin % 4 == 0, Cout % 4 == 0, cout_pad % 128 == 0; same geometry
 * arguments and the same reference call sites as ug_conv2d_f32. */
int ug_conv_split_weights(const float* w_packed, uint16_t* w_split, int taps, int Cin, int cout_pad, hipStream_t stream);
int ug_amax_f32(const float* x, int64_t rows, int64_t cols, int64_t ld, float* out_amax, hipStream_t stream);
/* the same into a slot the caller keeps zeroed (no memset ahead of the launch; ABI v5) */
int ug_amax_f32_into_zeroed(const float* x, int64_t rows, int64_t cols, int64_t ld, float* out_amax, hipStream_t stream);
int ug_conv2d_split(const float* x, const float* x_amax, const uint16_t* w_split, const float* bias, const float* residual,
                    float* y, int64_t B, int Hin, int Win, int Cin, int Cout, int cout_pad, int ksize, int stride,
                    int pad_top, int pad_left, int Hout, int Wout, int upsample2x, double* out_stats, int out_groups,
                    hipStream_t stream);   /* out_stats: as ug_conv3x3_split (needs Hout * Wout % 128 == 0), or null */
/* 3x3 / stride 1 / pad 1 convolution (every conv1/conv2 of ResnetBlock, conv_in/conv_out of the middle stacks:
 * common_modules.py:301-360, magvitv2.py:90-178) with the split operands of ug_conv2d_split and the 10 x 18 input
 * patch of an 8 x 16 output block resident in LDS for all nine taps.  With gn_mu_rstd (from ug_groupnorm_stats on
 * the same x) the load applies y = swish?(GroupNorm(x)) first -- the `conv(nonlinearity(norm(x)))` pattern of
 * ResnetBlock.forward -- so the normalised tensor never exists in HBM; zero padding applies to the normalised tensor,
 * as in the reference.  x_amax then bounds the NORMALISED tensor. */
int ug_conv3x3_split(const float* x, const float* x_amax, const uint16_t* w_split, const float* bias, const float* residual,
                     float* y, int64_t B, int H, int W, int Cin, int Cout, int cout_pad, const float* gn_mu_rstd,
                     const float* gn_gamma, const float* gn_beta, int gn_groups, int gn_swish, double* out_stats, int out_groups,
                     hipStream_t stream);
/* out_stats (or null): [B][out_groups][2] fp64 + one more 8-byte slot, ZEROED BY THE CALLER -- the sum and the sum of squares of
 * the stored y per (image, group), gathered in the convolution's epilogue with the arithmetic of ug_groupnorm_stats, for the
 * GroupNorm that consumes y (ResnetBlock.forward, common_modules.py:308-335: norm2 reads conv1's output, the next block's norm1
 * reads conv2's output + shortcut); ug_groupnorm_finalize turns them into (mean, rstd) without another pass over y.  The
 * trailing slot receives max|y| as an fp32 in its first four bytes (what ug_amax_f32 computes): the x_amax of a following split
 * convolution that has no GroupNorm on its load path (Downsample, nin_shortcut: common_modules.py:86-93,321-334). */
int ug_groupnorm_finalize(const double* stats, float* mu_rstd, int64_t B, int64_t HW, int C, int groups, float eps,
                          hipStream_t stream);
/* GroupNorm statistics only: stats_ws [B][groups][2] fp64 scratch, mu_rstd [B][groups][2] fp32 = (mean, rstd),
 * rounded as ug_groupnorm_swish rounds them (common_modules.py:19-27). */
int ug_groupnorm_stats(const float* x, double* stats_ws, float* mu_rstd, int64_t B, int64_t HW, int C, int groups,
                       float eps, hipStream_t stream);
/* replaces: SigLipAttention.forward (siglip_encoder.py:196-260): softmax(q k^T * scale) v per (image, head) in one
 * flash-style kernel, fp32 in / out, both contractions on the scaled two-way f16 split (fp32-accurate), scores never leave
 * registers.  qkv: rows (b*T + t), q | k | v blocks of H*head_dim columns, row stride ld; qkv_amax: device fp32 bound of
 * max|qkv| (ug_amax_f32) or null; out [B*T, H*head_dim] at row stride ldo.  head_dim % 4 == 0, <= 80. */
int ug_siglip_attn_f32(const float* qkv, int64_t ld, const float* qkv_amax, float* out, int64_t ldo, int64_t B, int64_t T,
                       int H, int head_dim, float scale, hipStream_t stream);
/* ug_linear_f32 on the same split-f16 contraction (SigLIP q/k/v/out_proj, fc1/fc2: siglip_encoder.py:196-199,
 * 250-259): y = act(x W^T + bias) + residual with W^T packed as ug_conv2d_f32 weights of a 1x1 conv ([1][K][n_pad])
 * and split by ug_conv_split_weights. */
int ug_linear_split(const float* x, int64_t ldx, const float* x_amax, const uint16_t* w_split, const float* bias,
                    const float* residual, int64_t ldres, float* y, int64_t ldy, int64_t M, int64_t N, int64_t K, int n_pad,
                    int act, hipStream_t stream);
/* batched fp32 GEMM on the same kernel (AttnBlock bmm's, common_modules.py:190-214) */
int ug_gemm_f32(const float* A, int64_t lda, int64_t stride_a, const float* B, int64_t ldb, int64_t stride_b,
                int b_is_nk, float* C, int64_t ldc, int64_t stride_c, int64_t M, int64_t N, int64_t K,
                int64_t batch, float alpha, hipStream_t stream);
/* ug_gemm_f32 over a two-level batch z = outer * batch_in + inner (operand z starts inner * s*_in + outer * s*_out
 * elements into its tensor): all heads of all images of SigLipAttention in one launch (siglip_encoder.py:196-230). */
int ug_gemm_f32_nested(const float* A, int64_t lda, int64_t sa_in, int64_t sa_out, const float* B, int64_t ldb,
                       int64_t sb_in, int64_t sb_out, int b_is_nk, float* C, int64_t ldc, int64_t sc_in, int64_t sc_out,
                       int64_t M, int64_t N, int64_t K, int64_t batch_in, int64_t batch_out, float alpha,
                       hipStream_t stream);
/* replaces: Normalize = GroupNorm(32, eps 1e-6) (+ swish), common_modules.py:19-27 */
int ug_groupnorm_swish(const float* x, const float* gamma, const float* beta, float* y, double* stats_ws,
                       int64_t B, int64_t HW, int C, int groups, float eps, int apply_swish, hipStream_t stream);
/* in-place softmax(scale * x) over the first `cols` columns of each row; columns [cols, min(ld, roundup(cols, 4))) are
 * set to zero so the probabilities can be contracted over a 16-byte-aligned width. */
int ug_softmax_rows_f32(float* x, int64_t rows, int64_t cols, int64_t ld, float scale, hipStream_t stream);
/* ---- SigLIP ViT (fp32, frozen; reference models/multimodal_encoder/siglip_encoder.py:152-309) ---- */
/* y = act(x W^T + bias) + residual, W [N][K] as nn.Linear stores it; act 1 = gelu_pytorch_tanh.
 * replaces: q/k/v/out_proj, fc1/fc2 (siglip_encoder.py:196-199,250-259); the 14x14/14 patch embedding is
 * ug_conv2d_f32 with ksize 14, stride 14. */
int ug_linear_f32(const float* x, int64_t ldx, const float* W, int64_t ldw, const float* bias,
                  const float* residual, int64_t ldres, float* y, int64_t ldy, int64_t M, int64_t N, int64_t K,
                  int act, hipStream_t stream);
/* replaces: nn.LayerNorm(eps 1e-6) (siglip_encoder.py:267-269) */
int ug_layernorm_f32(const float* x, const float* gamma, const float* beta, float* y, int64_t rows, int64_t cols,
                     float eps, hipStream_t stream);
/* Backward pieces of the tower for the UNFROZEN case (reference models/unigen.py:111 freeze=False; training/train_w_clip_vit.py:
 * 282,311-312).  The contractions of the backward run on ug_gemm_f32 / ug_gemm_f32_nested (exact fp32); these are the row-wise /
 * element-wise derivatives:
 *   ug_layernorm_bwd_f32      dx = [dres_in +] LayerNorm'(dy) (statistics recomputed from x); dgamma, dbeta ACCUMULATE
 *   ug_gelu_tanh_f32          dy_or_null == null: out = gelu_pytorch_tanh(pre); else out = dy * gelu'(pre)
 *   ug_softmax_bwd_rows_f32   in place on dP: dS = scale * P * (dP - sum_j dP_j P_j), padding columns [cols, ld) zeroed
 *   ug_colsum_f32             out[c] += sum_r x[r, c]   (bias / position-embedding gradients) */
int ug_layernorm_bwd_f32(const float* dy, const float* x, const float* gamma, const float* dres_in, float* dx, float* dgamma,
                         float* dbeta, int64_t rows, int64_t cols, float eps, hipStream_t stream);
int ug_gelu_tanh_f32(const float* pre, const float* dy_or_null, float* out, int64_t n, hipStream_t stream);
int ug_softmax_bwd_rows_f32(const float* P, float* dP, int64_t rows, int64_t cols, int64_t ld, float scale, hipStream_t stream);
int ug_colsum_f32(const float* x, int64_t ld, float* out, int64_t rows, int64_t cols, hipStream_t stream);
/* out[z][c][r] = in[z][r][c] for z < batch (operands z * stride apart); output rows ld_out >= rows long, their tail zero-filled:
 * the transposed activations / probabilities are the row-major A operand of the weight-gradient and dK / dV contractions */
int ug_transpose_f32(const float* in, int64_t ld_in, int64_t stride_in, float* out, int64_t ld_out, int64_t stride_out,
                     int64_t rows, int64_t cols, int64_t batch, hipStream_t stream);
int ug_nchw_to_nhwc(const float* in, float* out, int64_t B, int C, int64_t HW, int c_pad, hipStream_t stream);
int ug_nhwc_to_nchw(const float* in, float* out, int64_t B, int C, int64_t HW, int c_pad, hipStream_t stream);
/* replaces: LFQuantizer.get_indices / get_codebook_entry, magvitv2.py:210-230 */
int ug_lfq_pack(const float* z, int64_t ldz, int64_t* idx, int64_t n, int nbits, hipStream_t stream);
int ug_lfq_unpack(const int64_t* idx, float* z, int64_t n, int nbits, int* err_flag, hipStream_t stream);

/* ---- hardware probes (test-only: dump raw MFMA / LDS-transpose lane layouts) ---------------- */
int ug_probe_layouts(float* out, int64_t n_floats, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* UNIGEN_HIP_H */
