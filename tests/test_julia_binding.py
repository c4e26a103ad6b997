"""The reference-side binding (julia/LinearOperatorsMXLOExt.jl) against the C ABI it binds (CPU; no Julia needed).

Julia is not in this image, so the glue cannot be executed here. What CAN be checked mechanically is the part that
breaks silently at run time: every ``ccall((:name, lib|rccl), Ret, (ArgTypes...), args...)`` must name a function that
``include/mxlo.h`` / ``include/mxlo_rccl.h`` declare, in the right library, with the same arity, C-compatible
argument and return types, and as many actual arguments as declared types.
"""
import pathlib
import re

import pytest

ROOT = pathlib.Path(__file__).resolve().parents[1]
JL = ROOT / "julia" / "LinearOperatorsMXLOExt.jl"
HEADERS = {"lib": ROOT / "include" / "mxlo.h", "rccl": ROOT / "include" / "mxlo_rccl.h"}


# ---------------------------------------------------------------------------------------------- C side
def _strip_c_comments(src: str) -> str:
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    return re.sub(r"//[^\n]*", " ", src)


def _c_class(t: str) -> str:
    """Reduce a C parameter/return type to an ABI class: ptr | i32 | i64 | u64 | f64 | void."""
    t = t.strip()
    if "(*" in t:                                   # function pointer parameter written inline
        return "ptr"
    if "[" in t or "*" in t:
        return "ptr"
    t = re.sub(r"\bconst\b", "", t).strip()
    base = t.split()[0] if t.split() else ""
    if base == "void":
        return "void"
    if base in ("int32_t", "int"):
        return "i32"
    if base in ("int64_t", "long"):
        return "i64"
    if base in ("uint64_t", "size_t"):
        return "u64"
    if base == "double":
        return "f64"
    if base.endswith("_fn"):                        # typedef'd function pointer (mxlo_allreduce_fn)
        return "ptr"
    raise AssertionError(f"unclassified C type {t!r}")


def _split_top(s: str, sep: str = ",") -> list[str]:
    out, depth, cur = [], 0, []
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == sep and depth == 0:
            out.append("".join(cur).strip())
            cur = []
        else:
            cur.append(ch)
    last = "".join(cur).strip()
    if last:
        out.append(last)
    return out


def parse_header(path: pathlib.Path) -> dict[str, tuple[str, list[str]]]:
    src = _strip_c_comments(path.read_text())
    src = re.sub(r"^\s*#.*$", " ", src, flags=re.M)
    protos = {}
    # drop typedefs (struct bodies, function-pointer typedefs) so only prototypes remain
    src = re.sub(r"typedef\s+struct\s*\w*\s*\{.*?\}\s*\w+\s*;", " ", src, flags=re.S)
    src = re.sub(r"typedef[^;]*;", " ", src)
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(mxlo_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        ret, name, params = m.group(1), m.group(2), m.group(3)
        params = " ".join(params.split())
        plist = [] if params in ("", "void") else _split_top(params)
        protos[name] = (_c_class(ret if "*" in ret else ret.strip()), [_c_class(p_type(p)) for p in plist])
    return protos


def p_type(param: str) -> str:
    """'const void *v' -> 'const void *'; 'double scalars[5]' -> 'double []'."""
    param = param.strip()
    if "[" in param:
        return param[: param.index("[")].rsplit(None, 1)[0] + " []"
    m = re.match(r"(.*?)(\w+)$", param)
    if m and m.group(1).strip():
        return m.group(1)
    return param


# ---------------------------------------------------------------------------------------------- Julia side
def _jl_class(t: str) -> str:
    t = t.strip()
    if t == "P" or t.startswith("Ptr{") or t.startswith("Ref{") or t in ("Cstring", "Ptr"):
        return "ptr"
    return {"Int32": "i32", "Cint": "i32", "Int64": "i64", "UInt64": "u64", "Csize_t": "u64", "Float64": "f64",
            "Cdouble": "f64", "Cvoid": "void", "Nothing": "void"}.get(t) or pytest.fail(f"unclassified Julia type {t!r}")


def _strip_jl_comments(src: str) -> str:
    out = []
    for line in src.splitlines():
        in_str, k = False, 0
        while k < len(line):
            c = line[k]
            if c == '"' and (k == 0 or line[k - 1] != "\\"):
                in_str = not in_str
            elif c == "#" and not in_str:
                break
            k += 1
        out.append(line[:k])
    return "\n".join(out)


def parse_ccalls(path: pathlib.Path):
    src = _strip_jl_comments(path.read_text())
    calls = []
    for m in re.finditer(r"\bccall\(", src):
        i, depth = m.end(), 1
        while depth:
            ch = src[i]
            depth += ch in "([{"
            depth -= ch in ")]}"
            i += 1
        body = src[m.end(): i - 1]
        parts = _split_top(body)
        head = re.fullmatch(r"\(\s*:(\w+)\s*,\s*(\w+)\s*\)", parts[0])
        assert head, f"ccall target not of the form (:name, lib): {parts[0]!r}"
        types = parts[2].strip()
        assert types.startswith("(") and types.endswith(")"), parts[2]
        tl = _split_top(types[1:-1])
        line = src.count("\n", 0, m.start()) + 1
        calls.append({"name": head.group(1), "lib": head.group(2), "ret": parts[1].strip(), "types": tl,
                      "nargs": len(parts) - 3, "line": line})
    return calls


PROTOS = {k: parse_header(v) for k, v in HEADERS.items()}
CALLS = parse_ccalls(JL) + parse_ccalls(ROOT / "julia" / "runtests_mxlo.jl") + parse_ccalls(ROOT / "julia" / "bench.jl")


def test_headers_parse_to_the_full_symbol_lists():
    # sanity of the parser itself: a few prototypes with known shapes
    assert PROTOS["lib"]["mxlo_ctx_create"] == ("i32", ["i32", "ptr", "ptr"])
    assert PROTOS["lib"]["mxlo_version"] == ("ptr", [])
    assert PROTOS["lib"]["mxlo_qn_get_scalars"] == ("i32", ["ptr", "ptr", "ptr", "ptr"])
    assert PROTOS["lib"]["mxlo_ctx_set_allreduce"] == ("i32", ["ptr", "ptr", "ptr"])
    assert PROTOS["rccl"]["mxlo_shard_ctx_get"] == ("ptr", ["ptr", "i32"])
    assert len(PROTOS["lib"]) >= 65 and len(PROTOS["rccl"]) >= 25


def test_the_glue_has_ccalls_for_every_hot_path_leaf():
    names = {c["name"] for c in CALLS}
    must = {"mxlo_diag_mul", "mxlo_eye_mul", "mxlo_ones_mul", "mxlo_zeros_mul", "mxlo_householder_mul",
            "mxlo_hermitian_mul", "mxlo_gemv", "mxlo_gather", "mxlo_gather_range", "mxlo_scatter_zero_sorted",
            "mxlo_scatter_zero_range", "mxlo_blockdiag_create", "mxlo_blockdiag_mul", "mxlo_kron_mul",
            "mxlo_kron_mul_ex", "mxlo_qn_create", "mxlo_qn_push", "mxlo_qn_mul", "mxlo_qn_mul_shifted",
            "mxlo_qn_solve_shifted", "mxlo_qn_diag", "mxlo_qn_reset", "mxlo_diagqn_push", "mxlo_graph_begin",
            "mxlo_graph_end", "mxlo_graph_launch", "mxlo_diag_mul_c", "mxlo_eye_mul_c", "mxlo_zeros_mul_c",
            "mxlo_scale_c", "mxlo_conj_c", "mxlo_dot_c", "mxlo_householder_mul_c", "mxlo_gemv_c", "mxlo_hermitian_mul_c", "mxlo_kron_mul_c3", "mxlo_shard_ctx_create_ex",
            "mxlo_shard_ctx_preflight", "mxlo_index_plan_create", "mxlo_gather_plan", "mxlo_scatter_zero_plan",
            "mxlo_householder_mul_sharded", "mxlo_qn_create_sharded", "mxlo_qn_mul_sharded"}
    assert not (must - names), f"glue lacks ccalls for {sorted(must - names)}"


@pytest.mark.parametrize("call", CALLS, ids=[f"{c['name']}@{c['line']}" for c in CALLS])
def test_ccall_matches_header(call):
    where = f"julia/LinearOperatorsMXLOExt.jl:{call['line']} ccall(:{call['name']})"
    assert call["lib"] in PROTOS, f"{where}: unknown library constant {call['lib']}"
    protos = PROTOS[call["lib"]]
    other = PROTOS["rccl" if call["lib"] == "lib" else "lib"]
    assert call["name"] in protos, (
        f"{where}: not declared in {HEADERS[call['lib']].name}"
        + (" (it is declared in the OTHER library's header)" if call["name"] in other else ""))
    ret, params = protos[call["name"]]
    jl = [_jl_class(t) for t in call["types"]]
    assert len(jl) == len(params), f"{where}: {len(jl)} argument types, header declares {len(params)}"
    assert call["nargs"] == len(jl), f"{where}: {call['nargs']} actual arguments for {len(jl)} declared types"
    for k, (a, b) in enumerate(zip(jl, params)):
        assert a == b, f"{where}: argument {k + 1} is {call['types'][k]} ({a}), header says {b}"
    assert _jl_class(call["ret"]) == ret, f"{where}: return type {call['ret']}, header says {ret}"


def test_kernel_function_methods_are_defined_for_device_vectors():
    """The S-keyword hook: methods on the reference's kernel functions for MXVector arguments."""
    src = _strip_jl_comments(JL.read_text())
    for f in ("mulOpEye!", "mulOpOnes!", "mulOpZeros!", "mulSquareOpDiagonal!", "mulOpDiagonal!", "mulHouseholder!",
              "mulRestrict!", "multRestrict!"):
        assert re.search(rf"^(function\s+)?{re.escape(f)}\(res::MXVector", src, flags=re.M), f
        assert re.search(rf"import LinearOperators:[^#]*?{re.escape(f)}", src, flags=re.S), f"{f} not imported"
    assert re.search(r"^Base\.:\+\(op::LinearOperator\{T, MXVector\{T\}\}, x::Number\)", src, flags=re.M)


# ---------------------------------------------------------------------------------- the reference side of the binding
# Build-container check only (/root/reference does not travel to the GPU box): every name the extension imports from
# LinearOperators or qualifies with `LinearOperators.` must exist in the reference's sources, and every METHOD the
# extension adds to a reference function must have an arity the reference itself defines or calls — a renamed kernel
# function or a changed argument list upstream would otherwise only fail on a Julia-equipped host.
REF = pathlib.Path("/root/reference")
needs_ref = pytest.mark.skipif(not (REF / "src").is_dir(), reason="/root/reference is only present in the build container")


def _ref_sources() -> str:
    return "\n".join(_strip_jl_comments(p.read_text()) for p in sorted((REF / "src").glob("*.jl")))


def _imported_names() -> list[str]:
    src = _strip_jl_comments(JL.read_text())
    m = re.search(r"^import LinearOperators:(.*?)(?=^import |^using |^const )", src, flags=re.S | re.M)
    assert m, "import LinearOperators: ... block not found"
    return [n.strip() for n in m.group(1).replace("\n", " ").split(",") if n.strip()]


@needs_ref
def test_every_imported_reference_name_exists_in_the_reference():
    ref = _ref_sources()
    names = _imported_names()
    assert len(names) >= 25
    for n in names:
        pat = rf"(^|[^\w!]){re.escape(n)}(\{{[^}}]*\}})?\(|\b(struct|abstract type|mutable struct)\s+{re.escape(n)}\b|^\s*{re.escape(n)}\b.*="
        assert re.search(pat, ref, flags=re.M), f"`{n}` is imported by the extension but not defined in /root/reference/src"
    src = _strip_jl_comments("\n".join((ROOT / "julia" / f).read_text() for f in ("LinearOperatorsMXLOExt.jl", "runtests_mxlo.jl", "bench.jl")))
    for q in sorted(set(re.findall(r"\bLinearOperators\.(\w+!?)", src))):
        assert re.search(rf"(?<!\w){re.escape(q)}(?![\w!])", ref), f"LinearOperators.{q} is used by the glue but absent from the reference"


def _arities_in_reference(fname: str, ref: str) -> set[int]:
    """Positional argument counts of every definition AND call of `fname` in the reference sources."""
    out = set()
    for m in re.finditer(rf"(?<![\w.]){re.escape(fname)}(\{{[^}}]*\}})?\(", ref):
        i, depth = m.end(), 1
        while depth and i < len(ref):
            depth += ref[i] in "([{"
            depth -= ref[i] in ")]}"
            i += 1
        args = ref[m.end(): i - 1]
        pos = args.split(";")[0] if ";" in _split_top(args, ";")[0] or True else args
        pos = _split_top(args, ";")[0] if args.strip() else ""
        out.add(len([a for a in _split_top(pos) if a.strip()]))
    return out


@needs_ref
@pytest.mark.parametrize("fname", ["mulOpEye!", "mulOpOnes!", "mulOpZeros!", "mulSquareOpDiagonal!", "mulOpDiagonal!",
                                   "mulHouseholder!", "mulRestrict!", "multRestrict!", "opDiagonal", "opHouseholder",
                                   "opHermitian", "solve_shifted_system!", "diag!", "reset!", "storage_type",
                                   "has_args5", "isallocated5", "shifted_prod!"])
def test_methods_added_to_reference_functions_have_an_arity_the_reference_uses(fname):
    ref = _ref_sources()
    have = _arities_in_reference(fname, ref)
    assert have, f"{fname} does not occur in /root/reference/src"
    glue = _strip_jl_comments(JL.read_text())
    mine = set()
    for m in re.finditer(rf"^(?:function\s+)?(?:LinearOperators\.)?{re.escape(fname)}\(", glue, flags=re.M):
        i, depth = m.end(), 1
        while depth:
            depth += glue[i] in "([{"
            depth -= glue[i] in ")]}"
            i += 1
        args = glue[m.end(): i - 1]
        pos = _split_top(args, ";")[0] if args.strip() else ""
        mine.add(len([a for a in _split_top(pos) if a.strip()]))
    assert mine, f"the extension defines no method of {fname}"
    assert mine <= have, (f"{fname}: the extension defines methods with {sorted(mine)} positional arguments, the reference "
                          f"defines / calls it with {sorted(have)}")


def test_runtests_mirrors_the_reference_gpu_tests_line_by_line():
    """julia/runtests_mxlo.jl runs the reference's OWN storage-type test (test/gpu/test_S_kwarg.jl) by including that file
    and calling its `test_S_kwarg(arrayType = …)` with the extension's array types — the pattern of test/gpu/jlarrays.jl:1 /
    amdgpu.jl:2 (no re-typed assertion list) — and carries the assertions of test/gpu/amdgpu.jl:5-19 and the
    `@allocated == 0` block with the extension's types (checked textually; the Python mirror of the same assertions runs on
    the GPU in tests/test_gpu_callers.py::test_storage_type_kwarg_mirror)."""
    rt = (ROOT / "julia" / "runtests_mxlo.jl").read_text()
    code = "\n".join(ln for ln in rt.splitlines() if not ln.lstrip().startswith("#"))
    assert 'include(joinpath(pkgdir(LinearOperators), "test", "gpu", "test_S_kwarg.jl"))' in code
    assert "test_S_kwarg(arrayType = mxlo_array)" in code
    assert "LinearOperator(mat; S = vecTother)" not in code, "the S-kwarg assertions are the reference's: include its file, do not re-type them"
    for frag in ("BlockDiagonalOperator(A, B, C)", "y isa MXVector{Float32}", "storage_type(adjoint(A))", "storage_type(transpose(A))",
                 "storage_type(Diagonal(v)) == typeof(v)", "@allocated mul!(res, B, x)", "@allocated push!(HD, x, y, 1.0, x, tmpd)"):
        assert frag in rt, frag
    if (REF / "test" / "gpu" / "test_S_kwarg.jl").exists():       # what the include relies on still exists upstream
        up = (REF / "test" / "gpu" / "test_S_kwarg.jl").read_text()
        assert "function test_S_kwarg(; arrayType" in up
        for frag in ("LinearOperator(mat; S = vecTother)", "opRestriction([1, 2, 3], 32; S = vecT)", "arrayType(rand(Float32, 32, 32))",
                     "arrayType(rand(Float32, 32))"):
            assert frag in up, frag
        up2 = (REF / "test" / "gpu" / "amdgpu.jl").read_text()
        for frag in ("BlockDiagonalOperator(A, B, C)", "storage_type(Diagonal(v)) == typeof(v)"):
            assert frag in up2, frag
