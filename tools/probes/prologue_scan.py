"""Scalar-load round trips ahead of a kernel's first vector-memory instruction, from the device assembly:

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -Iml-unigen_amd/csrc -Iinclude -S --cuda-device-only ml-unigen_amd/csrc/decode.hip -o /tmp/decode.s
    python tools/probes/prologue_scan.py /tmp/decode.s

Per kernel: instructions before the first VMEM op, number of s_load, number of lgkmcnt waits, and the sequence (L = s_load, W = wait,
B = branch).  "LWBLLLW" = two DEPENDENT trips to the kernarg segment before the first vector load (~0.18 us each in a decode launch:
profiles/r06_decode_forms.md); one asm statement naming every argument (vmem_asm.h, argument hoisting) turns it into "LLLLW"."""
import sys,re
for path in sys.argv[1:]:
    L=open(path).read().split('\n')
    i=0
    while i<len(L):
        if L[i].startswith('_ZN') and '; @' in L[i]:
            name=L[i].split(':')[0]
            j=i+1; nins=0; waits=0; sloads=0; seq=[]
            while j<len(L) and 's_endpgm' not in L[j]:
                t=L[j].strip()
                if t and t[0] not in ';.' and not t.endswith(':'):
                    op=t.split()[0]; nins+=1
                    if op.startswith('s_load'): sloads+=1; seq.append('L')
                    if op=='s_waitcnt' and 'lgkmcnt' in t: waits+=1; seq.append('W')
                    if op.startswith('s_cbranch'): seq.append('B')
                    if op.startswith(('global_load','buffer_load','global_store','global_atomic')): break
                j+=1
            import subprocess
            dn=subprocess.run(['c++filt',name],capture_output=True,text=True).stdout.strip()[:90]
            print(f"{dn:92s} insts before first VMEM {nins:4d}  s_loads {sloads:2d}  lgkm waits {waits}  seq {''.join(seq)}")
        i+=1
