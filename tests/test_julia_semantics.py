"""reference == Julia glue == Python mirror, on SEMANTICS (round 6; VERDICT r5 "What's weak" #1).

tests/test_julia_binding.py checks that every `ccall` of julia/LinearOperatorsMXLOExt.jl matches the C headers. That net
let three behavioural drifts through (L-SR1 defaulting to scaling = false, a missing `push!(op, s, y, α, g)`, no redirect
of `push!(op, s, y)` on a damped operator). This module compares what the three sides SAY / DO about the things a caller
observes, with tests/golden/reference_semantics.json as the pivot:

* the JSON is re-derived from /root/reference/src whenever the reference is present (build container) and must be current;
* the glue's text is read with the same parser (tests/jl_semantics.py) and must state the same keywords + defaults, the
  same `push!` methods and — evaluating its guards over every (kind, damped, arity) — the same outcome per call, the same
  (symmetric, hermitian, tprod!, ctprod!) patterns, the same exception types;
* the Python mirror (the side the GPU suite executes) is checked by introspection and by running its `push` against a
  recording stand-in of the library on the CPU, and — GPU tests at the bottom — by building the live operators and making
  the real calls.

The three defects of round 5 are re-planted by text surgery in `test_the_checks_have_teeth_*` and must each fail.
"""
import inspect
import json
import pathlib
import re

import pytest
import torch

import jl_semantics as J

ROOT = pathlib.Path(__file__).resolve().parents[1]
GLUE = ROOT / "julia" / "LinearOperatorsMXLOExt.jl"
REF = pathlib.Path("/root/reference/src")
FACTS = json.loads((ROOT / "tests" / "golden" / "reference_semantics.json").read_text())
needs_ref = pytest.mark.skipif(not REF.is_dir(), reason="/root/reference is only present in the build container")


def glue_src() -> str:
    return J.strip_comments(GLUE.read_text())


# ------------------------------------------------------------------------------------------------ reference -> JSON
@needs_ref
def test_committed_reference_semantics_are_current():
    """tests/golden/reference_semantics.json is what tests/golden/make_semantics.py extracts from the reference NOW."""
    now = J.reference_facts(REF)
    committed = {k: v for k, v in FACTS.items() if not k.startswith("_")}
    assert json.loads(json.dumps(now)) == committed


def test_reference_semantics_say_what_the_reference_code_says():
    """Spot values read by hand in the reference (src/lsr1.jl:19, src/lbfgs.jl:26-35, :269-367, src/linalg.jl:94,118): the
    parser is not trusted blindly."""
    assert FACTS["qn_keywords"]["LSR1Data"] == {"mem": "5", "scaling": "true"}                 # NOT the docstring's false
    assert FACTS["qn_keywords"]["LBFGSData"] == {"mem": "5", "scaling": "true", "damped": "false", "inverse": "true",
                                                 "σ₂": "0.99", "σ₃": "10.0"}
    assert FACTS["push"]["arities"] == {"LBFGSOperator": [3, 4, 5, 6], "LSR1Operator": [3]}
    o = FACTS["push"]["outcomes"]
    assert o["fwd/damped=1/arity=3"] == "damped_fwd" and o["inv/damped=1/arity=3"] == "ErrorException"
    assert o["inv/damped=1/arity=5"] == "damped_inv" and o["fwd/damped=0/arity=4"] == "ErrorException"
    assert o["lsr1/damped=0/arity=4"] == "MethodError"
    assert FACTS["flags"]["opHouseholder [complex]"] == {"symmetric": False, "hermitian": True, "tprod": "nothing", "ctprod": "set"}
    assert FACTS["flags"]["opHermitian(d,A) [real]"] == {"symmetric": True, "hermitian": True, "tprod": "nothing", "ctprod": "nothing"}
    assert FACTS["flags"]["LSR1Operator [real]"] == {"symmetric": True, "hermitian": True, "tprod": "nothing", "ctprod": "nothing"}
    assert FACTS["errors"] == {"mul!: shape mismatch": "LinearOperatorException", "opHermitian: shape mismatch": "LinearOperatorException",
                               "solve_shifted_system!: σ < 0": "ArgumentError"}


# ------------------------------------------------------------------------------------------------------ glue vs JSON
def glue_keywords(src: str) -> dict:
    out = {}
    for name in ("mxqn", "lsr1", "ShardedQN"):
        fs = [f for f in J.functions(src, name) if f["kw"]]
        assert len(fs) == 1, f"{name}: {len(fs)} keyword methods in the glue"
        out[name] = fs[0]["kw"]
    return out


def check_glue_keywords(src: str):
    kw = glue_keywords(src)
    assert kw["mxqn"] == FACTS["qn_keywords"]["LBFGSData"], "mxqn(T, kind, n; ...) vs LBFGSData(T, n; ...) (src/lbfgs.jl:26-35)"
    assert kw["lsr1"] == FACTS["qn_keywords"]["LSR1Data"], "lsr1(T, n; ...) vs LSR1Data(T, n; ...) (src/lsr1.jl:19)"
    assert kw["ShardedQN"] == FACTS["qn_keywords"]["LBFGSData"], "ShardedQN(...; ...) keeps the constructor keywords"
    # the public constructors forward every keyword: L-BFGS kinds to mxqn, L-SR1 to lsr1 (which takes LSR1Data's two only)
    for name, target in (("InverseLBFGSOperator", r"mxqn\(T, 0, n; kw\.\.\.\)"), ("LBFGSOperator", r"mxqn\(T, 1, n; kw\.\.\.\)"),
                         ("LSR1Operator", r"lsr1\(T, n; kw\.\.\.\)")):
        fs = J.functions(src, name)
        assert len(fs) == 2, f"{name}: the (T, n, S) and the (n, S) method"
        for f in fs:
            assert list(f["kw"]) == ["kw..."] and re.search(target, f["body"]), f"{name} at glue line {f['line']}"
            assert "MXVector{T}" in f["pos"][-1], f"{name}: storage type as the last positional argument"


def test_glue_constructor_keywords_and_defaults_are_the_references():
    check_glue_keywords(glue_src())


def glue_push(src: str):
    return J.push_methods(src, r"MXQNOperator")


def check_glue_push(src: str):
    gm = glue_push(src)
    assert sorted(gm) == FACTS["push"]["arities"]["LBFGSOperator"], "push! methods on MXQNOperator vs src/lbfgs.jl:269-367"
    got = J.push_table(gm, gm)                    # one Julia type serves the three kinds: L-SR1 is guarded by op.kind == 2
    diff = {k: (FACTS["push"]["outcomes"][k], got[k]) for k in got if got[k] != FACTS["push"]["outcomes"][k]}
    assert not diff, f"(reference, glue) outcomes differ: {diff}"


def test_glue_push_methods_and_outcomes_are_the_references():
    check_glue_push(glue_src())


GLUE_FLAG_SITES = [   # (glue function, constructor regex, {(arity, real?, lsr1?): JSON key})
    ("opDiagonal", J.LINOP_CTOR, {(1, False, False): "opDiagonal(d) [complex]", (3, False, False): "opDiagonal(nrow,ncol,d) [complex,rectangular]"}),
    ("opHouseholder", J.LINOP_CTOR, {(1, True, False): "opHouseholder [real]", (1, False, False): "opHouseholder [complex]"}),
    ("opHermitian", J.LINOP_CTOR, {(2, True, False): "opHermitian(d,A) [real]", (2, False, False): "opHermitian(d,A) [complex]"}),
    ("mxqn", r"MXQNOperator\{[^}]*\}\(", {(3, True, False): "LBFGSOperator [real]", (3, True, True): "LSR1Operator [real]"}),
]


def test_glue_constructors_hand_over_the_references_flags():
    src = glue_src()
    seen = set()
    for fname, ctor, keys in GLUE_FLAG_SITES:
        for f in J.functions(src, fname):
            for real in J.method_scenarios(f):
                for lsr1 in ((False, True) if fname == "mxqn" else (False,)):
                    key = keys.get((len(f["pos"]), real, lsr1))
                    if key is None:
                        continue
                    env = {"real": real, "square": "rectangular" not in key, "lsr1": lsr1}
                    got = {tuple(sorted(J.flag_row(p).items())) for p in J.constructor_patterns(f["body"], ctor, env)}
                    assert got == {tuple(sorted(FACTS["flags"][key].items()))}, f"{fname} (glue line {f['line']}) vs {key}: {got}"
                    seen.add(key)
    assert seen == {k for _, _, keys in GLUE_FLAG_SITES for k in keys.values()}
    assert FACTS["flags"]["InverseLBFGSOperator [real]"] == FACTS["flags"]["LBFGSOperator [real]"]     # one glue constructor serves both


def test_glue_maps_statuses_to_the_references_exception_types():
    src = glue_src()
    chk = [f for f in J.functions(src, "check") if f["pos"] == ["st::Int32"]][0]["body"]
    assert re.search(r"st == 2 && throw\(LinearOperatorException\(", chk)       # MXLO_ESHAPE
    assert re.search(r"st == 6 && throw\(ArgumentError\(", chk)                  # MXLO_EDOMAIN: σ < 0
    assert re.search(r"\n\s*error\(lasterr\(\)\)", chk)                          # everything else: ErrorException
    assert FACTS["errors"]["mul!: shape mismatch"] == "LinearOperatorException"
    assert FACTS["errors"]["solve_shifted_system!: σ < 0"] == "ArgumentError"
    for f in J.functions(src, "opHermitian"):
        assert re.search(r'\|\| throw\(LinearOperatorException\("shape mismatch"\)\)', f["body"]), f["line"]
    hdr = (ROOT / "include" / "mxlo.h").read_text()
    assert re.search(r"#define MXLO_ESHAPE\s+2\b", hdr) and re.search(r"#define MXLO_EDOMAIN\s+6\b", hdr)


@needs_ref
def test_integration_md_table_is_current():
    """INTEGRATION.md §3a "reference line -> glue line -> mirror line" is what tools/semantics_table.py prints for THIS tree."""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, str(ROOT / "tools" / "semantics_table.py")], capture_output=True, text=True, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    doc = (ROOT / "INTEGRATION.md").read_text()
    rows = [ln for ln in out.stdout.splitlines() if ln.startswith("| ")]
    assert len(rows) >= 12
    stale = [ln for ln in rows if ln not in doc]
    assert not stale, "re-run tools/semantics_table.py and paste its table into INTEGRATION.md §3a:\n" + "\n".join(stale)


# -------------------------------------------------------------------------------------------- the checks have teeth
def test_the_checks_have_teeth_lsr1_default_scaling():
    """Round 5's defect (a): `scaling::Bool = kind != 2` made L-SR1 default to scaling = false."""
    bad = glue_src().replace("function mxqn(::Type{T}, kind::Integer, n::Int; mem::Int = 5, scaling::Bool = true",
                             "function mxqn(::Type{T}, kind::Integer, n::Int; mem::Int = 5, scaling::Bool = kind != 2")
    assert bad != glue_src()
    with pytest.raises(AssertionError, match="LBFGSData"):
        check_glue_keywords(bad)
    bad = glue_src().replace("lsr1(::Type{T}, n::Int; mem::Int = 5, scaling::Bool = true)", "lsr1(::Type{T}, n::Int; mem::Int = 5, scaling::Bool = false)")
    assert bad != glue_src()
    with pytest.raises(AssertionError, match="LSR1Data"):
        check_glue_keywords(bad)


def test_the_checks_have_teeth_missing_push_method_and_redirect():
    """Round 5's defects (b): no `push!(op, s, y, α, g)`; `push!(op, s, y)` on a damped operator went to the plain update."""
    src = glue_src()
    five = [f for f in J.functions(src, "push!") if len(f["pos"]) == 5 and "MXQNOperator" in f["pos"][0]]
    assert len(five) == 1
    bad = src.replace(five[0]["body"], "\n  op\n")          # method body gone: no redirect to the 6-argument form
    bad = re.sub(r"function push!\(op::MXQNOperator\{T\}, s::MXVector\{T\}, y::MXVector\{T\}, α::T, g::MXVector\{T\}\) where \{T\}",
                 "function push5!(op::MXQNOperator{T}, s::MXVector{T}, y::MXVector{T}, α::T, g::MXVector{T}) where {T}", bad)
    with pytest.raises(AssertionError, match="push! methods on MXQNOperator"):
        check_glue_push(bad)
    bad = re.sub(r"  if op\.data\.damped[^\n]*\n    return push!\(op, s, y, similar\(s\)\)\n  end\n", "", src)
    assert bad != src
    with pytest.raises(AssertionError, match=r"fwd/damped=1/arity=3"):
        check_glue_push(bad)


# ---------------------------------------------------------------------------------------------------- mirror vs JSON
JL_LITERAL = {"true": True, "false": False}


def jl_value(text: str):
    return JL_LITERAL[text] if text in JL_LITERAL else (int(text) if re.fullmatch(r"-?\d+", text) else float(text))


def test_mirror_constructor_keywords_and_defaults_are_the_references(lo):
    from linearoperators_jl_amd import qn
    spelled = {"σ₂": "sigma2", "σ₃": "sigma3"}
    for ctor in (lo.InverseLBFGSOperator, lo.LBFGSOperator):
        sig = inspect.signature(ctor)
        for name, default in FACTS["qn_keywords"]["LBFGSData"].items():
            if name == "inverse":                               # accepted and ignored, as `delete!(kwargs, :inverse)` (lbfgs.jl:115,171)
                continue
            p = sig.parameters[spelled.get(name, name)]
            assert p.default == jl_value(default), (ctor.__name__, name)
        assert any(p.kind == p.VAR_KEYWORD for p in sig.parameters.values()), "σ₂ / σ₃ / inverse by their reference spelling"
    assert qn._kw({"inverse": False, "σ₂": 0.5}, 0.99, 10.0) == (0.5, 10.0)
    with pytest.raises(TypeError):
        qn._kw({"scalling": True}, 0.99, 10.0)                  # a keyword typo is an error, as in Julia
    sig = inspect.signature(lo.LSR1Operator)
    assert {n for n in sig.parameters} - {"T", "n", "device"} == set(FACTS["qn_keywords"]["LSR1Data"])
    for name, default in FACTS["qn_keywords"]["LSR1Data"].items():
        assert sig.parameters[name].default == jl_value(default), name


ENTRY = {"mxlo_qn_push": "plain", "mxlo_qn_push_damped_fwd": "damped_fwd", "mxlo_qn_push_damped_inv": "damped_inv"}


def mirror_push_outcome(lo, op, n, dev, arity, calls):
    """One mirror `push` of the given arity; the outcome in the reference's vocabulary."""
    mk = lambda: torch.ones(n, dtype=torch.float64, device=dev)
    args = {3: (), 4: (mk(),), 5: (0.5, mk()), 6: (0.5, mk(), mk())}[arity]
    del calls[:]
    try:
        lo.push(op, mk(), mk() * 2.0, *args)
    except RuntimeError as e:
        assert "status" not in str(e), f"the mirror must refuse BEFORE the library does: {e}"
        return "ErrorException"
    except TypeError:
        return "MethodError"
    pushes = [c for c in calls if c in ENTRY]
    assert len(pushes) == 1, calls
    return ENTRY[pushes[0]]


def test_mirror_push_outcomes_with_a_recording_library(lo, monkeypatch):
    """The mirror's `push` run on the CPU against a stand-in that records which entry point it would call."""
    from linearoperators_jl_amd import _lib, qn
    calls = []
    monkeypatch.setattr(_lib, "call", lambda name, *a: calls.append(name))
    monkeypatch.setattr(qn, "check_vec", lambda t, name, dtype=None: t)
    monkeypatch.setattr(qn, "ptr", lambda t: 0)

    class Ctx:
        def bind_stream(self):
            pass

    n = 8
    got = {}
    for kind, damped, arity in J.PUSH_CASES:
        cls = qn.LSR1OperatorType if kind == "lsr1" else qn.LBFGSOperatorType
        op = object.__new__(cls)
        op.eltype, op.nrow, op.ncol, op._ctx, op._h = torch.float64, n, n, Ctx(), None
        op.damped, op.inverse, op._nupdate = damped, kind == "inv", 0
        got[f"{kind}/damped={int(damped)}/arity={arity}"] = mirror_push_outcome(lo, op, n, "cpu", arity, calls)
    diff = {k: (FACTS["push"]["outcomes"][k], got[k]) for k in got if got[k] != FACTS["push"]["outcomes"][k]}
    assert not diff, f"(reference, mirror) outcomes differ: {diff}"


# ------------------------------------------------------------------------------------------- mirror, live (GPU box)
@pytest.mark.gpu
def test_mirror_push_outcomes_on_the_device(lo, dev, monkeypatch):
    """The same table with REAL operators and the real library: every (kind, damped, arity) ends where the reference's does."""
    from linearoperators_jl_amd import _lib
    calls, real = [], _lib.call
    monkeypatch.setattr(_lib, "call", lambda name, *a: (calls.append(name), real(name, *a))[1])
    n = 300
    ctor = {"fwd": lo.LBFGSOperator, "inv": lo.InverseLBFGSOperator}
    got = {}
    for kind, damped, arity in J.PUSH_CASES:
        op = lo.LSR1Operator(n, mem=3, device=dev) if kind == "lsr1" else ctor[kind](n, mem=3, damped=damped, device=dev)
        got[f"{kind}/damped={int(damped)}/arity={arity}"] = mirror_push_outcome(lo, op, n, dev, arity, calls)
    assert got == FACTS["push"]["outcomes"]


@pytest.mark.gpu
def test_mirror_operators_carry_the_references_flags(lo, dev):
    """(symmetric, hermitian, tprod! is nothing, ctprod! is nothing) of the live mirror operators vs the reference's
    constructor calls (tests/golden/reference_semantics.json "flags")."""
    f64, c128 = torch.float64, torch.complex128
    vec = lambda n, dt=f64: torch.ones(n, dtype=dt, device=dev)
    S = lambda dt: lo.Storage(dt, dev)
    A = torch.ones(6, 6, dtype=f64, device=dev).t()
    Ac = torch.ones(6, 6, dtype=c128, device=dev).t()
    built = {
        "opEye/square [real]": lo.opEye(f64, 5, S=S(f64)),
        "opEye/rectangular [real,rectangular]": lo.opEye(f64, 5, 7, S=S(f64)),
        "opOnes [real]": lo.opOnes(f64, 5, 5, S=S(f64)),
        "opOnes [real,rectangular]": lo.opOnes(f64, 5, 7, S=S(f64)),
        "opZeros [real]": lo.opZeros(f64, 5, 5, S=S(f64)),
        "opZeros [real,rectangular]": lo.opZeros(f64, 5, 7, S=S(f64)),
        "opDiagonal(d) [real]": lo.opDiagonal(vec(5)),
        "opDiagonal(d) [complex]": lo.opDiagonal(vec(5, c128)),
        "opDiagonal(nrow,ncol,d) [real,rectangular]": lo.opDiagonal(5, 7, vec(5)),
        "opDiagonal(nrow,ncol,d) [complex,rectangular]": lo.opDiagonal(5, 7, vec(5, c128)),
        "opRestriction [real,rectangular]": lo.opRestriction([1, 3], 5, device=dev),
        "opHouseholder [real]": lo.opHouseholder(vec(5)),
        "opHouseholder [complex]": lo.opHouseholder(vec(5, c128)),
        "opHermitian(d,A) [real]": lo.opHermitian(vec(6), A),
        "opHermitian(d,A) [complex]": lo.opHermitian(vec(6), Ac),
        "InverseLBFGSOperator [real]": lo.InverseLBFGSOperator(9, device=dev),
        "LBFGSOperator [real]": lo.LBFGSOperator(9, device=dev),
        "LSR1Operator [real]": lo.LSR1Operator(9, device=dev),
        "hcat [real,rectangular]": lo.hcat(lo.opEye(f64, 5, S=S(f64)), lo.opOnes(f64, 5, 2, S=S(f64))),
        "vcat [real,rectangular]": lo.vcat(lo.opEye(f64, 5, S=S(f64)), lo.opOnes(f64, 2, 5, S=S(f64))),
    }
    assert set(built) == set(FACTS["flags"]), "one live operator per reference constructor row"
    for key, op in built.items():
        got = {"symmetric": bool(op.symmetric), "hermitian": bool(op.hermitian),
               "tprod": "nothing" if op.tprod is None else "set", "ctprod": "nothing" if op.ctprod is None else "set"}
        assert got == FACTS["flags"][key], key
    # the scalar defaults a caller never passes: read back from the handles
    assert lo.LSR1Operator(9, device=dev).scaling is True and lo.LBFGSOperator(9, device=dev).scaling is True
    assert lo.LSR1Operator(9, device=dev).mem == 5 and lo.LBFGSOperator(9, device=dev).damped is False
