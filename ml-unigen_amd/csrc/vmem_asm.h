// Hand-issued vector-memory operations with counted waits (gfx950).
//
// Why: with the builtins (`__builtin_amdgcn_global_load_lds`, plain loads) hipcc's wait insertion treats ANY LDS access as a possible
// alias of every LDS-DMA in flight and puts `s_waitcnt vmcnt(0)` in front of it.  In the decode kernels that meant: the operand image
// (an LDS write) was built only after the WHOLE weight ring had landed, and the attention prologue only after the K / V chunk -- the
// "4.5 us per launch that are not weight streaming" of rounds 4-5 (ISA evidence: profiles/r06_vmcnt_findings.md).  Issued through asm
// volatile the compiler sees no VMEM at all: ordering = program order of the asm statements, data readiness = the waits written here.
// Rules: (1) every wait counts only LOADS issued behind the one waited for -- stores / atomics in between are never counted, so a wait
// can over-wait but not under-wait whatever order stores retire in; (2) a loaded value is used only behind tie() placed after its wait.
#pragma once
#include <stdint.h>

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// the same for a count that only becomes a constant after unrolling (the switch folds away)
__device__ __forceinline__ void wait_vm_n(int n) {
#define UG_W(v) case v: asm volatile("s_waitcnt vmcnt(" #v ")" ::: "memory"); break;
  switch (n) {
    UG_W(0) UG_W(1) UG_W(2) UG_W(3) UG_W(4) UG_W(6) UG_W(8) UG_W(9) UG_W(12) UG_W(16) UG_W(24) UG_W(32) UG_W(40) UG_W(48)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
#undef UG_W
}
// 16 bytes per lane HBM -> LDS (lane-linear at lds_addr); scalar base + 32-bit lane offset.  nt: a byte read once by one CU.
__device__ __forceinline__ void dma16_nt(uint64_t base, uint32_t voff, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" ::"v"(voff), "s"(base), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ void dma16(uint64_t base, uint32_t voff, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds_addr) : "memory");
}
// the same with a per-lane 64-bit address
__device__ __forceinline__ void dma16_nt(const void* src, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt" ::"v"(src), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ void dma16(const void* src, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds_addr) : "memory");
}
template <int OFF, typename T>
__device__ __forceinline__ void ld16(T& v, uint64_t base, uint32_t voff) {
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=&v"(v) : "v"(voff), "s"(base), "n"(OFF) : "memory");
}
template <typename T>
__device__ __forceinline__ void ld16(T& v, const void* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(v) : "v"(p) : "memory"); }
template <typename T>
__device__ __forceinline__ void ld8(T& v, const void* p) { asm volatile("global_load_dwordx2 %0, %1, off" : "=&v"(v) : "v"(p) : "memory"); }
template <typename T>
__device__ __forceinline__ void ld4(T& v, const void* p) { asm volatile("global_load_dword %0, %1, off" : "=&v"(v) : "v"(p) : "memory"); }
template <typename T>
__device__ __forceinline__ void ld2u(T& v, const void* p) { asm volatile("global_load_ushort %0, %1, off" : "=&v"(v) : "v"(p) : "memory"); }
// a value loaded above may only be used behind the wait that covers it: re-define it there
template <typename T>
__device__ __forceinline__ void tie(T& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ uint32_t lds_addr_of(const void* p) { return (uint32_t)(uintptr_t)p; }      // low half of a generic LDS pointer
// Kernel arguments ("argument hoisting").  hipcc fetches a kernel argument by a scalar load where it is first needed; behind an early
// exit or a data-dependent branch that is a second, third ... DEPENDENT round trip to the kernarg segment before the kernel's first
// vector load (0.18 us each; a decode launch lives ~4 us; ISA scan: tools/probes/prologue_scan.py).  The decode kernels therefore
//   (1) take the pointers / sizes their first loads need as LEADING SCALAR arguments: built with -mllvm -amdgpu-kernarg-preload-count=16
//       (Makefile) the first 14 dwords arrive in SGPRs with the wave (a by-value struct is passed by reference and never preloaded);
//   (2) name every other argument in ONE statement  asm volatile("" :: "s"(arg), "s"(arg), ...);  placed behind the first loads: the
//       operands must be live SGPRs there, so all of them are fetched in one batch behind one wait (several statements would each end
//       a scheduling region and bring the dependent trips back; implicit arguments -- gridDim, blockDim -- count too).
// A wave-uniform word ANOTHER launch wrote (the decode position): hipcc may not treat it as constant, so it reads it with a VECTOR load
// and waits vmcnt(0) for it on the spot -- a serialised L2 round trip at the top of the kernel.  Hand-issued scalar load instead (the
// scalar cache is invalidated at every kernel start); use the value behind wait_lgkm0() + tie_s().
__device__ __forceinline__ void sld4(int& v, const void* p) { asm volatile("s_load_dword %0, %1, 0x0" : "=&s"(v) : "s"(p) : "memory"); }
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
template <typename T>
__device__ __forceinline__ void tie_s(T& v) { asm volatile("" : "+s"(v)); }

