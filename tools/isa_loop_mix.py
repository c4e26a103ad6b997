#!/usr/bin/env python
"""Instruction mix of the kron GEMM's steady-state loop, read from the gfx950 ISA (no GPU needed: hipcc cross-compiles).
For every gemm_glds_kernel instantiation in csrc/dense.hip: the basic block with the most MFMAs and its counts of MFMA /
LDS / DMA / scalar / vector-ALU / waitcnt / barrier instructions, plus the scratch size of the kernel.
  python tools/isa_loop_mix.py > profiles/r0N_gemm_isa_mix.txt"""
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "linearoperators.jl_amd", "csrc")


def cls(ins):
    op = ins.split()[0]
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("global_load_lds"):
        return "dma"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("v_"):
        return "valu"
    return "other"


def main():
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "dense.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++20", "--offload-arch=gfx950", "-ffp-contract=off",
                        "--cuda-device-only", "-S", "dense.hip", "-o", out], cwd=CSRC, check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        L = open(out).read().split("\n")
    scratch = {}
    name = None
    for l in L:
        m = re.match(r"\s+\.name:\s+(\S+)", l)
        if m:
            name = m.group(1)
        m = re.match(r"\s+\.private_segment_fixed_size:\s+(\d+)", l)
        if m and name:
            scratch[name] = int(m.group(1))
    print("kernel<T,CA,CB | BETA0 AK | TM TN WM WN BK NST | PAIR PFD SWAPC NTC XNOBAR UNR>: steady-loop block -> counts per trip\n")
    seen = set()
    for st, l in enumerate(L):
        if not (l.startswith("_ZN4mxlo16gemm_glds_kernelI") and "@" in l):
            continue
        sym = l.split(":")[0]
        end = st
        while not L[end].startswith(".Lfunc_end"):
            end += 1
        blocks, cur = [], None
        for x in L[st:end]:
            if re.match(r"^\.LBB\d+_\d+:", x):
                cur = [x.split(":")[0], []]
                blocks.append(cur)
            elif cur is not None and x.startswith("\t") and not x.strip().startswith((".", ";")):
                cur[1].append(x.strip())
        b = max(blocks, key=lambda b: sum(1 for i in b[1] if i.startswith("v_mfma")))
        c = Counter(cls(i) for i in b[1])
        m = re.match(r"_ZN4mxlo16gemm_glds_kernelI(\w)(\w)(\w)Lb(\d)ELb(\d)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb\dELb\dELb(\d)ELi(\d)ELb(\d)ELb(\d)ELb(\d)ELb(\d)E", sym)
        if not m:
            continue
        g = m.groups()
        key = (g[0], g[4]) + g[5:]           # element type, A layout, tile configuration and options
        if key in seen:                        # the scalar-type / beta instantiations share the loop
            continue
        seen.add(key)
        valu = Counter(i.split()[0] for i in b[1] if cls(i) == "valu").most_common(3)
        print(f"{'f64' if g[0] == 'd' else 'f32'} {'AK' if g[4] == '1' else 'AM'} t{g[5]}x{g[6]} w{g[7]}x{g[8]} bk{g[9]} st{g[10]}"
              f" pair{g[11]} pfd{g[12]} swapc{g[13]} unr{g[16]}: mfma {c['mfma']:3d}  lds {c['lds']:3d}  dma {c['dma']:2d}  salu {c['salu']:3d}"
              f"  valu {c['valu']:2d}  waitcnt {c['waitcnt']:2d}  barrier {c['barrier']}  scratch {scratch.get(sym, '?')} B"
              + (f"   (valu: {valu})" if valu else ""))


if __name__ == "__main__":
    sys.exit(main())
