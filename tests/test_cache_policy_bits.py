"""CPU: the cache-policy hints are IN the compiled kernels. Round 6 found that `flag ? __builtin_nontemporal_load(p) : *p` with a
run-time flag makes hipcc merge the two loads and drop the `nt` bit from both, silently (opHermitian ran 8 % slower at n = 16384 and
nothing failed). The policy is a template parameter since; this test disassembles the gfx950 code objects out of the in-tree
object files and checks that the NT = true instantiations of the opHermitian pass kernels issue their 16 tile loads per tile with
`nt`, the NT = false ones without."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "linearoperators.jl_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"


def _disassemble(obj, tmp):
    fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
    subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True)
    subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                    f"--input={fat}", f"--output={co}"], check=True)
    out = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout
    funcs, name = {}, None
    for line in out.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:$", line)
        if m:
            name = m.group(1)
            funcs[name] = []
        elif name is not None:
            funcs[name].append(line)
    return funcs


@pytest.mark.parametrize("obj,pattern", [("dense.o", r"herm_pass(_block)?_kernelI[df]Li\d+E(Li\d+E)?Lb([01])E"),
                                         ("complex.o", r"cherm_pass_kernelI[df]Li\d+ELb([01])E")])
def test_nontemporal_hint_is_in_the_hermitian_pass_kernels(tmp_path, obj, pattern):
    path = os.path.join(CSRC, obj)
    if not (os.path.exists(path) and shutil.which("objcopy") and os.path.exists(os.path.join(LLVM, "llvm-objdump"))):
        pytest.skip("needs the in-tree object files and the ROCm LLVM tools")
    funcs = _disassemble(path, str(tmp_path))
    seen = {"0": 0, "1": 0}
    for name, body in funcs.items():
        m = re.search(pattern, name)
        if not m:
            continue
        nt = m.groups()[-1]
        loads = [ln for ln in body if "global_load_dwordx4" in ln]
        with_nt = [ln for ln in loads if re.search(r"\bnt\b", ln)]
        assert len(loads) >= 16, (name, len(loads))
        if nt == "1":
            assert len(with_nt) >= 16 and len(with_nt) % 16 == 0, (name, len(loads), len(with_nt))   # (f32: the slice of v is a plain dwordx4 load)
        else:
            assert not with_nt, (name, len(with_nt))
        seen[nt] += 1
    assert seen["0"] >= 3 and seen["1"] >= 3, seen
