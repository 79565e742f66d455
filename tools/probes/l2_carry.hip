// Does a weight slice touched by launch N stay in the XCD's L2 (or the Infinity Cache) for launch N+1?  (round 5, AR decode:
// "keep HBM streaming across launch boundaries".)  Build: hipcc --offload-arch=gfx950 -O3 l2_carry.hip -o _build/l2_carry
//
// consumer C: G workgroups x W waves, every wave pulls T tiles of 8 KB by LDS-DMA (8 x 1 KB, nt) exactly like the decode GEMVs and
//             leaves; its duration is taken INSIDE the kernel (s_memrealtime: min start .. max end over the workgroups).
// toucher  T: P workgroups; workgroup p walks the slices of the consumer workgroups b with (b + shift) % 8 == p % 8 and loads one
//             dword per 128-byte line (default cache policy).  shift = 0: the same XCD that will consume the slice (L2 candidate);
//             shift = 1: a different XCD (the slice can only come back from the Infinity Cache).
// Buffers rotate over > 600 MB so that nothing is resident unless a toucher put it there.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Rec { unsigned long long t0, t1; int xcc, pad; };

__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15; }

template <int W, int T>
__global__ __launch_bounds__(64 * W) void consume(const char* __restrict__ buf, Rec* __restrict__ rec, int nt_policy) {
  __shared__ __attribute__((aligned(1024))) char tile[W][T > 1 ? 2 : 1][8192];       // two-slot ring like the GEMVs
  const unsigned long long t0 = wall_clock64();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const char* src = buf + ((size_t)blockIdx.x * W + wave) * T * 8192;
#pragma unroll
  for (int t = 0; t < T; ++t) {
    if (t >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");          // tile t - 2 has landed: its slot is free
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (nt_policy) __builtin_amdgcn_global_load_lds((gptr_t)(src + t * 8192 + i * 1024 + lane * 16), (lptr_t)(tile[wave][t & 1] + i * 1024), 16, 0, 2);
      else __builtin_amdgcn_global_load_lds((gptr_t)(src + t * 8192 + i * 1024 + lane * 16), (lptr_t)(tile[wave][t & 1] + i * 1024), 16, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    rec[blockIdx.x].t0 = t0; rec[blockIdx.x].t1 = wall_clock64(); rec[blockIdx.x].xcc = xcc_id();
    if (tile[0][0][5] == 77 && tile[W - 1][0][9] == 78) rec[blockIdx.x].pad = 1;      // keeps the LDS image alive
  }
}

// The same stream with the REAL weight layout: W [N][K] row-major bf16, a tile = 16 rows x 512 bytes at a row stride of 2 K bytes
// (workgroup b -> chunk b / nslabs, slab b % nslabs; wave w -> row groups (chunk W + w) T + t), against the packed form above in
// which a tile is 8 KB of consecutive bytes.
template <int W, int T>
__global__ __launch_bounds__(64 * W) void consume_strided(const char* __restrict__ buf, Rec* __restrict__ rec, int K, int nslabs, int N) {
  __shared__ __attribute__((aligned(1024))) char tile[W][T > 1 ? 2 : 1][8192];
  const unsigned long long t0 = wall_clock64();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int chunk = blockIdx.x / nslabs, slab = blockIdx.x % nslabs;
#pragma unroll
  for (int t = 0; t < T; ++t) {
    if (t >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    const int grp = (chunk * W + wave) * T + t;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = min(grp * 16 + 2 * i + (lane >> 5), N - 1);
      const char* src = buf + ((size_t)row * K + slab * 256 + (lane & 31) * 8) * 2;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(tile[wave][t & 1] + i * 1024), 16, 0, 2);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    rec[blockIdx.x].t0 = t0; rec[blockIdx.x].t1 = wall_clock64(); rec[blockIdx.x].xcc = xcc_id();
    if (tile[0][0][5] == 77 && tile[W - 1][0][9] == 78) rec[blockIdx.x].pad = 1;
  }
}

// consumer workgroup slice = W * T * 8 KB; toucher workgroup p (256 threads) serves consumer workgroups b = p%8 - shift + 8 j (mod
// placement), j = p/8, p/8 + P/8, ...
__global__ __launch_bounds__(256) void touch(const char* __restrict__ buf, int G, size_t slice, int shift, Rec* __restrict__ rec, int* sink) {
  const unsigned long long t0 = wall_clock64();
  const int P8 = gridDim.x >> 3, xcd = blockIdx.x & 7, j0 = blockIdx.x >> 3;
  int acc = 0;
  for (int j = j0; j * 8 < G + 8; j += P8) {
    const int b = j * 8 + ((xcd + 8 - shift) & 7);
    if (b >= G) continue;
    const char* s = buf + (size_t)b * slice;
#pragma unroll 8
    for (size_t off = (size_t)threadIdx.x * 128; off < slice; off += 256 * 128) acc ^= *reinterpret_cast<const int*>(s + off);
  }
  if (acc == 0x12345679) *sink = acc;
  __syncthreads();
  if (threadIdx.x == 0) { rec[blockIdx.x].t0 = t0; rec[blockIdx.x].t1 = wall_clock64(); rec[blockIdx.x].xcc = xcc_id(); }
}

// an unrelated stream between toucher and consumer (what the attention launch or another GEMV does to the caches)
typedef float f4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void stream_other(const f4_t* __restrict__ src, size_t n4, float* sink) {
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const f4_t v = __builtin_nontemporal_load(src + i);
    s += v.x;
  }
  if (s == 1.2345f) *sink = s;
}

template <int W, int T>
void run_case(const char* name, int G, hipStream_t st) {
  const size_t slice = (size_t)W * T * 8192, bytes = slice * G;
  const int NB = (int)std::max<size_t>(6, (size_t)700e6 / bytes + 1);
  std::vector<char*> bufs(NB);
  for (auto& b : bufs) { CK(hipMalloc(&b, bytes)); CK(hipMemset(b, 1, bytes)); }
  char* other; const size_t other_bytes = 8u << 20; CK(hipMalloc(&other, other_bytes * 8)); CK(hipMemset(other, 0, other_bytes * 8));
  const int P = 64;
  Rec *rc, *rt; CK(hipMalloc(&rc, sizeof(Rec) * G * NB)); CK(hipMalloc(&rt, sizeof(Rec) * P * NB));
  int* sink; CK(hipMalloc(&sink, 64));
  std::vector<Rec> hc(G * NB), ht(P * NB);
  auto span = [&](std::vector<Rec>& h, int n, int i) {
    unsigned long long a = ~0ull, b = 0;
    for (int k = 0; k < n; ++k) { a = std::min(a, h[i * n + k].t0); b = std::max(b, h[i * n + k].t1); }
    return (double)(b - a) * 0.01;       // 100 MHz -> us
  };
  // mode: 0 cold, 1 touch same XCD, 2 touch other XCD, 3 touch same XCD + 8 MB unrelated stream in between, 4 same buffer every time
  //       5 touch same XCD, consumer with the default cache policy instead of nt
  printf("%-8s %6.2f MB  G=%d x %d waves x %d tiles, %d buffers\n", name, bytes / 1e6, G, W, T, NB);
  for (int mode = 0; mode < 6; ++mode) {
    double best_c = 1e9, best_t = 1e9, mean_c = 0; int nm = 0; int bad_xcc = 0;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemsetAsync(rc, 0, sizeof(Rec) * G * NB, st));
      for (int i = 0; i < NB; ++i) {
        char* b = mode == 4 ? bufs[0] : bufs[i];
        if (mode == 1 || mode == 2 || mode == 3 || mode == 5)
          hipLaunchKernelGGL(touch, dim3(P), dim3(256), 0, st, b, G, slice, mode == 2 ? 1 : 0, rt + i * P, sink);
        if (mode == 3) hipLaunchKernelGGL(stream_other, dim3(256), dim3(256), 0, st, (const f4_t*)(other + (i % 8) * other_bytes), other_bytes / 16, (float*)sink);
        hipLaunchKernelGGL((consume<W, T>), dim3(G), dim3(64 * W), 0, st, b, rc + i * G, mode == 5 ? 0 : 1);
      }
      CK(hipStreamSynchronize(st));
      CK(hipMemcpy(hc.data(), rc, sizeof(Rec) * G * NB, hipMemcpyDeviceToHost));
      CK(hipMemcpy(ht.data(), rt, sizeof(Rec) * P * NB, hipMemcpyDeviceToHost));
      for (int i = (mode == 4 ? 2 : 0); i < NB; ++i) {
        const double c = span(hc, G, i);
        best_c = std::min(best_c, c); mean_c += c; ++nm;
        if (mode == 1 || mode == 2 || mode == 3 || mode == 5) best_t = std::min(best_t, span(ht, P, i));
      }
      for (int k = 0; k < G; ++k) if (hc[k].xcc != (k & 7)) ++bad_xcc;
    }
    static const char* mn[] = {"cold", "touched by the same XCD", "touched by another XCD (Infinity Cache only)",
                               "same XCD + 8 MB unrelated stream between", "same buffer every launch", "same XCD, consumer default policy"};
    printf("  %-48s consumer %6.2f us best / %6.2f mean (%5.2f TB/s best)", mn[mode], best_c, mean_c / nm, bytes / best_c / 1e6);
    if (best_t < 1e8) printf("   toucher %6.2f us", best_t);
    printf("   [workgroups off XCD b%%8: %d]\n", bad_xcc);
  }
  for (auto b : bufs) CK(hipFree(b));
  CK(hipFree(other)); CK(hipFree(rc)); CK(hipFree(rt)); CK(hipFree(sink));
}

template <int W, int T>
void run_layout(const char* name, int N, int K, hipStream_t st) {
  const int nslabs = K / 256, groups = N / 16, G = ((groups + W * T - 1) / (W * T)) * nslabs;
  const size_t bytes = (size_t)N * K * 2 + 8192 * W * T;
  const int NB = (int)std::max<size_t>(6, (size_t)700e6 / bytes + 1);
  std::vector<char*> bufs(NB);
  for (auto& b : bufs) { CK(hipMalloc(&b, bytes)); CK(hipMemset(b, 1, bytes)); }
  Rec* rc; CK(hipMalloc(&rc, sizeof(Rec) * G * NB));
  std::vector<Rec> hc(G * NB);
  printf("%-8s N=%d K=%d  %.2f MB  G=%d x %d waves x %d tiles\n", name, N, K, (double)N * K * 2 / 1e6, G, W, T);
  for (int layout = 0; layout < 2; ++layout) {
    double best = 1e9, mean = 0; int nm = 0;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float wall = 0;
    for (int rep = 0; rep < 4; ++rep) {
      hipEventRecord(e0, st);
      for (int i = 0; i < NB; ++i) {
        if (layout == 0) hipLaunchKernelGGL((consume<W, T>), dim3(G), dim3(64 * W), 0, st, bufs[i], rc + i * G, 1);
        else hipLaunchKernelGGL((consume_strided<W, T>), dim3(G), dim3(64 * W), 0, st, bufs[i], rc + i * G, K, nslabs, N);
      }
      hipEventRecord(e1, st);
      CK(hipStreamSynchronize(st));
      hipEventElapsedTime(&wall, e0, e1);
      CK(hipMemcpy(hc.data(), rc, sizeof(Rec) * G * NB, hipMemcpyDeviceToHost));
      for (int i = 0; i < NB; ++i) {
        unsigned long long a = ~0ull, b = 0;
        for (int k = 0; k < G; ++k) { a = std::min(a, hc[i * G + k].t0); b = std::max(b, hc[i * G + k].t1); }
        const double c = (double)(b - a) * 0.01;
        best = std::min(best, c); mean += c; ++nm;
      }
    }
    printf("  %-44s in-kernel %6.2f us best / %6.2f mean (%5.2f TB/s mean);  wall per launch %6.2f us\n",
           layout == 0 ? "packed: a tile = 8 KB contiguous" : "row-major [N][K]: 16 x 512 B at stride 2K", best, mean / nm,
           (double)N * K * 2 / (mean / nm) / 1e6, wall * 1e3 / NB);
  }
  for (auto b : bufs) CK(hipFree(b));
  CK(hipFree(rc));
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  run_layout<9, 3>("gate_up", 17920, 1536, st);
  run_layout<8, 2>("down", 1536, 8960, st);
  run_layout<4, 1>("qkv", 2048, 1536, st);
  run_case<1, 1>("o", 576, st);            // 576 one-wave workgroups x 8 KB  = 4.7 MB   (gemv_ring_kernel<1,1>)
  run_case<4, 1>("qkv", 192, st);          // 192 x 4 waves x 8 KB            = 6.3 MB   (gemv_ring4<1,1,RESID_NORM,4>)
  run_case<8, 2>("down", 210, st);         // 210 x 8 x 2                     = 27.5 MB  (gemv_ring4<1,2,SWIGLU,8>)
  run_case<9, 3>("gate_up", 249, st);      // 249 x 9 x 3                     = 55 MB    (gemv_ring4<1,3,RESID_NORM,9>: ring of two slots there)
  return 0;
}
