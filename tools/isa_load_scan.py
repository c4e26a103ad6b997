#!/usr/bin/env python
"""How many global loads does a kernel keep in flight? Compiles one .hip of csrc/ to gfx950 assembly (device side only,
no GPU needed) and reports, per kernel with at least MIN_LOADS global loads: the number of loads, the number of
`s_waitcnt vmcnt(0)` (a full drain), and the longest run of loads issued without a full drain in between. A kernel
whose longest run is 1-2 while it has many loads waits for (almost) every load before issuing the next one: fine where
8-10 waves per SIMD hide the latency, a chain of memory round trips at launch-bound sizes (round 4: the push pass).
  python tools/isa_load_scan.py reductions.hip [name-substring ...] > profiles/r0N_isa_load_scan_<file>.txt"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "linearoperators.jl_amd", "csrc")
MIN_LOADS = 6


def main():
    src = os.path.join(CSRC, sys.argv[1])
    want = sys.argv[2:]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++20", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off",
                        "-fvisibility=hidden", "--cuda-device-only", "-S", src, "-o", out], check=True,
                       stderr=subprocess.DEVNULL)
        txt = open(out).read()
    parts = re.split(r"\n(_Z\w+):\s*; @", txt)
    rows = []
    for i in range(1, len(parts), 2):
        body = parts[i + 1].split("s_endpgm")[0]
        run = best = loads = drains = 0
        for line in body.splitlines():
            if "global_load" in line or "buffer_load" in line:
                loads += 1
                run += 1
                best = max(best, run)
            elif "s_waitcnt vmcnt(0)" in line:
                drains += 1
                run = 0
        if loads >= MIN_LOADS:
            rows.append((parts[i], loads, drains, best))
    names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.splitlines()
    print(f"# {sys.argv[1]}: kernels with >= {MIN_LOADS} global loads; longest run of loads without a full vmcnt drain")
    print(f"{'loads':>5s} {'drains':>6s} {'run':>4s}  kernel")
    for (m, loads, drains, best), name in sorted(zip(rows, names), key=lambda t: (t[0][3], t[1])):
        if want and not any(w in name for w in want):
            continue
        print(f"{loads:5d} {drains:6d} {best:4d}  {name[:170]}")


if __name__ == "__main__":
    main()
