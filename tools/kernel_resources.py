#!/usr/bin/env python
"""Register / LDS / scratch usage of every kernel in the built libmxlo.so, read from the gfx950 code objects inside the
HIP fat binary (no GPU needed). Prints the kernels matching the given substrings (default: the hot ones), one line each:
VGPRs (+AGPRs), SGPRs, static LDS bytes, scratch bytes, and the waves per SIMD the VGPR count allows (512 / vgprs).
  python tools/kernel_resources.py [substring ...] > profiles/r0N_kernel_resources.txt"""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "linearoperators.jl_amd", "csrc", "libmxlo.so")
LLVM = "/opt/rocm/lib/llvm/bin"
HOT = ["map_kernel", "panel_dots_kernel", "push_pass_kernel", "combine_kernel", "panel_gemm_kernel", "gemm_glds_kernel",
       "herm_pass_kernel", "herm_finish_kernel", "blockdiag_kernel", "householder_fused_kernel", "qn_apply_fused_kernel",
       "gather_idx_kernel", "extend_sorted_kernel", "gemvb_", "gemv_"]


def kernels():
    blob = open(SO, "rb").read()
    out = []
    with tempfile.TemporaryDirectory() as td:
        for bi, m in enumerate(re.finditer(re.escape(b"__CLANG_OFFLOAD_BUNDLE__"), blob)):
            p = m.start()
            (n,) = struct.unpack_from("<Q", blob, p + 24)
            off = p + 32
            for e in range(n):
                o, size, tl = struct.unpack_from("<QQQ", blob, off)
                off += 24
                triple = blob[off:off + tl].decode()
                off += tl
                if "gfx950" not in triple or size == 0:
                    continue
                f = os.path.join(td, f"co_{bi}_{e}.co")
                with open(f, "wb") as fh:
                    fh.write(blob[p + o:p + o + size])
                notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", f], capture_output=True, text=True, check=True).stdout
                cur = {}
                for line in notes.splitlines():
                    mm = re.match(r"\s+-?\s*\.(\w+):\s+(\S+)", line)
                    if not mm:
                        continue
                    k, v = mm.groups()
                    if k == "agpr_count" and "name" in cur and "vgpr_count" in cur:   # a new kernel record begins with .agpr_count
                        out.append(cur)
                        cur = {}
                    cur[k] = v
                if "name" in cur:
                    out.append(cur)
    return out


def main():
    pats = sys.argv[1:] or HOT
    ks = [k for k in kernels() if "name" in k and "vgpr_count" in k]
    names = [k["name"] for k in ks]
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    rows = []
    for k, d in zip(ks, dem):
        if not any(p in d for p in pats):
            continue
        d = re.sub(r"\(anonymous namespace\)::|mxlo::|void ", "", d)
        d = re.sub(r"\(.*$", "", d)
        v, a = int(k.get("vgpr_count", 0)), int(k.get("agpr_count", 0))
        rows.append((d, v, a, int(k.get("sgpr_count", 0)), int(k.get("group_segment_fixed_size", 0)),
                     int(k.get("private_segment_fixed_size", 0))))
    rows = sorted(set(rows))
    print(f"{len(ks)} kernels in libmxlo.so; {sum(1 for k in ks if int(k.get('private_segment_fixed_size', 0)) > 0)} use scratch\n")
    print(f"{'kernel':110s} vgpr agpr sgpr    lds scratch waves/SIMD")
    for d, v, a, s, l, sc in rows:
        tot = max(v + a, 1)
        print(f"{d[:110]:110s} {v:4d} {a:4d} {s:4d} {l:6d} {sc:7d} {min(8, 512 // ((tot + 7) // 8 * 8)):3d}")


if __name__ == "__main__":
    main()
