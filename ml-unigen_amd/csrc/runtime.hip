// Library plumbing: thread-local error string, ABI version, and a hardware probe used by the GPU
// tests to record raw lane layouts (LDS transpose-read) next to the kernels that will rely on them.
#include "common.h"
#include "unigen_hip.h"
#include <string.h>
#include <new>

static thread_local char g_err[512] = "";

void ug_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* ug_last_error(void) { return g_err; }
extern "C" int ug_abi_version(void) { return UG_ABI_VERSION; }

extern "C" int ug_create(ug_handle** out) {
  UG_REQUIRE(out != nullptr, "ug_create: null output pointer");
  ug_handle* h = new (std::nothrow) ug_handle;
  UG_REQUIRE(h != nullptr, "ug_create: out of host memory");
  h->tail_ws = nullptr;
  const size_t bytes = (size_t)UG_HANDLE_WS_SLOTS * 256 * 256 * sizeof(float);
  // private partials are fully written before they are read: the scratch needs no clearing
  if (hipMalloc(&h->tail_ws, bytes) != hipSuccess) {
    (void)hipGetLastError();
    delete h;
    ug_set_error("ug_create: hipMalloc of the %zu-byte GEMM scratch failed", bytes);
    return UG_ERR_STATE;
  }
  *out = h;
  return UG_OK;
}

extern "C" int ug_destroy(ug_handle* h) {
  if (!h) return UG_OK;
  if (h->tail_ws) (void)hipFree(h->tail_ws);
  delete h;
  return UG_OK;
}

namespace {
typedef __attribute__((ext_vector_type(4))) short s16x4;
// out[lane*4 + j] = element j returned to `lane` by ds_read_b64_tr_b16 when LDS holds lds[i] = i
// (16-bit) and lane reads at byte address lane*8.   out[256 + lane*4 + r] = C-layout probe of
// mfma_f32_16x16x32_bf16 with A[i][k] = (k==0 ? i : 0), B[k][j] = (k==0 ? 1 : 0) * (j+1) -> D[i][j] = i*(j+1)
__global__ void probe_kernel(float* out) {
  __shared__ __attribute__((aligned(16))) short lds[1024];
  const int lane = threadIdx.x;
  for (int i = lane; i < 1024; i += 64) lds[i] = (short)i;
  __syncthreads();
  s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + lane * 4));
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (float)t[j];
  bf16x8_t a, b;
  for (int s = 0; s < 8; ++s) { a[s] = 0; b[s] = 0; }
  if ((lane >> 4) == 0) { a[0] = (short)f2bf((float)(lane & 15)); b[0] = (short)f2bf((float)((lane & 15) + 1)); }
  f32x4_t c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[256 + lane * 4 + r] = c[r];
}
}  // namespace

extern "C" int ug_probe_layouts(float* out, int64_t n_floats, hipStream_t st) {
  UG_REQUIRE(out && n_floats >= 512, "ug_probe_layouts: need >= 512 floats");
  hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, st, out);
  UG_CHECK_LAUNCH("ug_probe_layouts");
  return UG_OK;
}
