// Device-side assembly of the t2i training sequence (reference UniversalPromptingQwen2.t2i_prompt,
// training/prompting_utils.py:59-111, without the random prompt dropout which stays a host decision):
//   [pad ... | conv_start  TEXT  conv_end | <soi>  image tokens  <eoi>]      left-padded to max_seq_len,
// text truncated on the right when it does not fit; labels = ignore on the prompt, <soi>/<eoi> ids at their own
// positions, the MaskGIT labels on the image positions; any label equal to pad_id becomes ignore.
#include "common.h"
#include "unigen_hip.h"

namespace {

__global__ __launch_bounds__(256) void t2i_assemble_kernel(const int64_t* __restrict__ text, const int64_t* __restrict__ offs,
                                                           const int64_t* __restrict__ conv_start, int ns,
                                                           const int64_t* __restrict__ conv_end, int ne,
                                                           const int64_t* __restrict__ image_in, const int64_t* __restrict__ image_lab,
                                                           int n, int L, int64_t pad_id, int64_t soi_id, int64_t eoi_id,
                                                           int64_t ignore_id, int64_t* __restrict__ ids, int64_t* __restrict__ labels,
                                                           uint8_t* __restrict__ attn01) {
  const int b = blockIdx.x;
  const int64_t t0 = offs[b];
  const int tlen = (int)(offs[b + 1] - t0);
  const int room = L - n - 2;
  const int body = ns + tlen + ne;
  const int npad = room >= body ? room - body : 0;
  for (int p = threadIdx.x; p < L; p += blockDim.x) {
    int64_t id, lab;
    if (p < room) {
      if (p < npad) id = pad_id;
      else {
        const int j = p - npad;                                   // index into conv_start | text | conv_end
        id = j < ns ? conv_start[j] : j < ns + tlen ? text[t0 + j - ns] : conv_end[j - ns - tlen];
      }
      lab = ignore_id;
    } else if (p == room) { id = soi_id; lab = soi_id; }
    else if (p == L - 1) { id = eoi_id; lab = eoi_id; }
    else { id = image_in[(int64_t)b * n + p - room - 1]; lab = image_lab[(int64_t)b * n + p - room - 1]; }
    if (lab == pad_id) lab = ignore_id;
    ids[(int64_t)b * L + p] = id;
    labels[(int64_t)b * L + p] = lab;
    if (attn01) attn01[(int64_t)b * L + p] = p >= npad;
  }
}

// MaskGIT training-time masking (data/masking.py:13-94, default branch).  The reference draws scores ~ U(0,1) [B, n],
// takes `perm = scores.argsort(-1)` and masks position j iff perm[j] < k_b (k_b = round(n * mask_prob_b) clamped to >= 1):
// position j is masked iff the element with the j-th smallest score has an index below k_b.  With rank(i) = number of
// scores ordered before element i (ties broken by index, i.e. a stable sort), that is mask[rank(i)] = 1 for every
// i < k_b -- no sort needed.  One workgroup per row, the row's scores in LDS.
__global__ __launch_bounds__(256) void maskgit_train_mask_kernel(const int64_t* __restrict__ tokens, const float* __restrict__ scores,
                                                                 const float* __restrict__ num_masked, int n, int64_t mask_id,
                                                                 int64_t ignore_id, int64_t* __restrict__ input_ids,
                                                                 int64_t* __restrict__ labels) {
  extern __shared__ float lds[];
  float* s = lds;                                       // n scores
  unsigned char* m = reinterpret_cast<unsigned char*>(lds + n);   // n mask bytes
  const int b = blockIdx.x;
  for (int i = threadIdx.x; i < n; i += blockDim.x) { s[i] = scores[(int64_t)b * n + i]; m[i] = 0; }
  __syncthreads();
  const int k = (int)num_masked[b];
  for (int i = threadIdx.x; i < k && i < n; i += blockDim.x) {
    const float si = s[i];
    int r = 0;
    for (int j = 0; j < n; ++j) r += (s[j] < si) || (s[j] == si && j < i);
    m[r] = 1;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int64_t t = tokens[(int64_t)b * n + i];
    input_ids[(int64_t)b * n + i] = m[i] ? mask_id : t;
    labels[(int64_t)b * n + i] = m[i] ? t : ignore_id;
  }
}

}  // namespace

extern "C" int ug_t2i_assemble(const int64_t* text_ids, const int64_t* text_offsets, const int64_t* conv_start, int64_t n_start,
                               const int64_t* conv_end, int64_t n_end, const int64_t* image_in, const int64_t* image_labels,
                               int64_t B, int64_t n_image, int64_t max_seq_len, int64_t pad_id, int64_t soi_id, int64_t eoi_id,
                               int64_t ignore_id, int64_t* input_ids, int64_t* labels, uint8_t* attn01, hipStream_t st) {
  UG_REQUIRE(text_ids && text_offsets && image_in && image_labels && input_ids && labels && B > 0 && n_image > 0 &&
                 max_seq_len >= n_image + 2 && n_start >= 0 && n_end >= 0 && (n_start == 0 || conv_start) && (n_end == 0 || conv_end),
             "ug_t2i_assemble: bad args (B=%ld n=%ld L=%ld)", (long)B, (long)n_image, (long)max_seq_len);
  hipLaunchKernelGGL(t2i_assemble_kernel, dim3((unsigned)B), dim3(256), 0, st, text_ids, text_offsets, conv_start, (int)n_start,
                     conv_end, (int)n_end, image_in, image_labels, (int)n_image, (int)max_seq_len, pad_id, soi_id, eoi_id, ignore_id,
                     input_ids, labels, attn01);
  UG_CHECK_LAUNCH("ug_t2i_assemble");
  return UG_OK;
}

extern "C" int ug_maskgit_train_mask(const int64_t* tokens, const float* scores, const float* num_masked, int64_t B, int64_t n,
                                     int64_t mask_id, int64_t ignore_id, int64_t* input_ids, int64_t* labels, hipStream_t st) {
  UG_REQUIRE(tokens && scores && num_masked && input_ids && labels && B > 0 && n > 0 && n <= 8192,
             "ug_maskgit_train_mask: bad args (B=%ld n=%ld, n <= 8192)", (long)B, (long)n);
  const size_t lds = (size_t)n * sizeof(float) + (size_t)n;
  hipLaunchKernelGGL(maskgit_train_mask_kernel, dim3((unsigned)B), dim3(256), lds, st, tokens, scores, num_masked, (int)n, mask_id,
                     ignore_id, input_ids, labels);
  UG_CHECK_LAUNCH("ug_maskgit_train_mask");
  return UG_OK;
}
