"""Probe build of the library whose fp32-operand decode GEMV (gemv_ring4_kernel) stamps s_memrealtime at seven points per wave:

    python tools/probes/decode_trace_build.py   ->  tools/probes/_build/libunigen_hip_dtrace.so   (+ ug_decode_trace_set)

The product source stays free of the instrumentation: it is inserted into a copy of decode.hip at the anchors below.  Stamps:
0 kernel entry | 1 operand loads + two weight tiles issued | 2 operand loads landed | 3 operand image in LDS (after the
workgroup barrier) | 4 first weight tile landed | 5 every MFMA done, every atomic issued | 6 atomics acknowledged (vmcnt 0).
Regions: 0 = gate/up (RESID_NORM, 9 waves), 1 = q/k/v (RESID_NORM, 4 waves), 2 = down (SWIGLU).  Reader: decode_trace.py."""
import os
import subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src_dir = os.path.join(ROOT, "ml-unigen_amd", "csrc")
out = os.path.join(ROOT, "tools", "probes", "_build")
os.makedirs(out, exist_ok=True)
s = open(os.path.join(src_dir, "decode.hip")).read()


def sub(old, new, count=1):
    global s
    assert s.count(old) == count, (s.count(old), old)
    s = s.replace(old, new)


sub("constexpr int DHD = 128;\n", """constexpr int DHD = 128;
__device__ unsigned long long* g_trace = nullptr;
// stamps stay in registers (s_memtime, the shader clock) and are written once at the kernel's end; slot 7 = s_memrealtime at entry
#define UG_STAMP(k) ug_st[k] = clock64()
""")
sub("""  constexpr int UPW = (8 + NW - 1) / NW;           // k-steps of the operand each wave converts
""", """  constexpr int UPW = (8 + NW - 1) / NW;           // k-steps of the operand each wave converts
  unsigned long long* const ug_tr = g_trace;
  unsigned long long ug_st[8];
  ug_st[7] = wall_clock64();
  UG_STAMP(0);
""")
sub("""  stage(0);
  if constexpr (KW > 1) stage(1);
  decode_clear(f, threadIdx.x, linear_block());
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const int ar = min(rb * 16 + row, R - 1);
    const bool live""", """  stage(0);
  if constexpr (KW > 1) stage(1);
  UG_STAMP(1);
  if constexpr (KW > 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  UG_STAMP(2);
  decode_clear(f, threadIdx.x, linear_block());
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const int ar = min(rb * 16 + row, R - 1);
    const bool live""")
sub("""  lds_barrier();
  bf16x8_t xf[RB][8];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int u = 0; u < 8; ++u) xf[rb][u] = frag[rb][u][lane];
#pragma unroll
  for (int t = 0; t < KW; ++t) {
    if (t + 1 < KW) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const char* tr = tile[wave][t & 1] + row * 512;""", """  lds_barrier();
  UG_STAMP(3);
  bf16x8_t xf[RB][8];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int u = 0; u < 8; ++u) xf[rb][u] = frag[rb][u][lane];
#pragma unroll
  for (int t = 0; t < KW; ++t) {
    if (t + 1 < KW) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (t == 0) UG_STAMP(4);
    const char* tr = tile[wave][t & 1] + row * 512;""")
sub("""        if (r < R && n < N) atomicAdd(ap + __umul24(r, sr), d[rb][j]);
      }
    }
  }
}


template <int RB>
void launch_gemv(""", """        if (r < R && n < N) atomicAdd(ap + __umul24(r, sr), d[rb][j]);
      }
    }
  }
  UG_STAMP(5);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  UG_STAMP(6);
  if (ug_tr && (threadIdx.x & 63) == 0) {
    unsigned long long* o = ug_tr + ((XIN == XIN_RESID_NORM ? (NW == 9 ? 0 : 1) : 2) * 4096 + linear_block() * 16 + (threadIdx.x >> 6)) * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = ug_st[k];
  }
}


template <int RB>
void launch_gemv(""")
sub("""extern "C" int ug_gemv_bf16(""", """extern "C" int ug_decode_trace_set(unsigned long long* p) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &p, sizeof(p)) == hipSuccess ? UG_OK : UG_ERR_LAUNCH;
}

extern "C" int ug_gemv_bf16(""")
# the constexpr UPW line follows nmain only in ring4; make sure the anchor existed
tmp = os.path.join(out, "decode_trace.hip")
open(tmp, "w").write(s)
obj = os.path.join(out, "decode_trace.o")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-I" + src_dir,
                       "-I" + os.path.join(ROOT, "include"), "-c", tmp, "-o", obj])
objs = [obj if f == "decode.hip" else os.path.join(src_dir, f.replace(".hip", ".o")) for f in sorted(os.listdir(src_dir)) if f.endswith(".hip")]
so = os.path.join(out, "libunigen_hip_dtrace.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-ldl", "-o", so])
os.remove(obj)
print(so)
