#!/usr/bin/env python3
"""Instruction census of the MFMA main loops in the built library (no GPU needed):

    python tools/loop_census.py halo2_kernelI4BF16 wgrad9_kernelI4BF16 ...      (substrings of mangled kernel names)

Extracts the gfx950 code objects of streamyolo_amd/lib/libstreamyolo_hip.so, finds in every matching kernel the innermost
backward branch whose body holds the most MFMAs and prints instructions per class and "others per MFMA".  The guide's rule of
thumb (MI355X_MICROARCH.md, MFMA table): at one wave per SIMD at most ~5 single-issue instructions hide behind one
v_mfma_f32_32x32x16_bf16 (32 cycles = 8 issue slots); loops above that are issue-bound whatever the memory system does."""
import os, shutil, tempfile
import re, sys, glob, subprocess, collections
pat = sys.argv[1:]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORK = tempfile.mkdtemp(prefix='sy_census_')
shutil.copy(os.path.join(ROOT, 'streamyolo_amd', 'lib', 'libstreamyolo_hip.so'), WORK)
subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-objdump', '--offloading', 'libstreamyolo_hip.so'], cwd=WORK, capture_output=True)
def classify(l):
    op=l.split()[0]
    if op.startswith('v_mfma'): return 'mfma'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('buffer_','global_','flat_','scratch_')): return 'lds_dma' if l.rstrip().endswith('lds') else 'vmem'
    if op=='s_waitcnt': return 'wait'
    if op=='s_barrier': return 'barrier'
    if op.startswith('s_'): return 'salu'
    if op.startswith('v_'): return 'valu'
    return 'other'
for f in sorted(glob.glob(WORK + '/*.hipv4-amdgcn-amd-amdhsa--gfx950')):
    txt=subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump","-d",f],capture_output=True,text=True).stdout
    cur=None; ins=[]
    def flush():
        if cur is None or not any(p in cur for p in pat): return
        # loops: backward branches
        best=None
        for i,(a,l) in enumerate(ins):
            m=re.match(r's_cbranch_\w+\s+(\d+)',l) or re.match(r's_branch\s+(\d+)',l)
            if not m: continue
            off=int(m.group(1)); off = off-65536 if off>=32768 else off
            if off>=0: continue
            tgt=a+4+off*4
            body=[x for x in ins if tgt<=x[0]<=a]
            n=sum(1 for x in body if x[1].startswith('v_mfma'))
            if n and (best is None or n>best[0]): best=(n,body)
        if best:
            n,body=best
            c=collections.Counter(classify(l) for a,l in body)
            # bytes per wave-instruction (64 lanes): what one MFMA slot asks of LDS and of the L1 / L2 path
            width={'b128':1024,'dwordx4':1024,'b96':768,'dwordx3':768,'b64':512,'dwordx2':512,'b32':256,'dword':256}
            def nbytes(l):
                op=l.split()[0]
                for k,v in width.items():
                    if op.endswith(k) or ('_'+k+'_') in op or op.endswith(k+'_tr_b16') or (k in op and op.startswith('ds_read')): return v
                return 0
            lds_b=sum(nbytes(l) for a,l in body if l.split()[0].startswith('ds_read'))
            l2_b=sum(nbytes(l) for a,l in body if l.split()[0].startswith(('buffer_load','global_load')))
            print("%-78s loop %4d instr, %3d mfma -> %.2f others/mfma, LDS reads %.2f KB/mfma, global loads %.2f KB/mfma  %s" %
                  (cur[14:92], len(body), n, (len(body)-n)/n, lds_b/n/1024, l2_b/n/1024, {k:v for k,v in sorted(c.items()) if k!='mfma'}))
    for line in txt.splitlines():
        m=re.match(r'^[0-9a-f]+ <(\S+)>:',line)
        if m: flush(); cur=m.group(1); ins=[]; continue
        m=re.match(r'^\s+(\S.*?)\s*//\s*([0-9A-Fa-f]+):',line)
        if m: ins.append((int(m.group(2),16), m.group(1)))
    flush()
shutil.rmtree(WORK, ignore_errors=True)
