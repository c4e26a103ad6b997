#!/usr/bin/env python
"""Prints the "reference line -> glue line -> mirror line" table of INTEGRATION.md §3a with CURRENT line numbers
(build container only: reads /root/reference/src). Facts come from tests/jl_semantics.py, the same parser
tests/test_julia_semantics.py uses."""
import inspect
import os
import pathlib
import re
import sys

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import jl_semantics as J  # noqa: E402
import __graft_entry__ as g  # noqa: E402

lo = g.load_package()
from linearoperators_jl_amd import leaves, qn  # noqa: E402

REF = pathlib.Path("/root/reference/src")
ref = {p.name: J.strip_comments(p.read_text()) for p in REF.glob("*.jl")}
glue = J.strip_comments((ROOT / "julia" / "LinearOperatorsMXLOExt.jl").read_text())


def pyline(obj, pattern=None):
    src, start = inspect.getsourcelines(obj)
    if pattern is None:
        return start
    for k, ln in enumerate(src):
        if re.search(pattern, ln):
            return start + k
    raise SystemExit(f"{pattern!r} not found in {obj}")


def jl(srcs, file, name, arity=None, typ=None):
    fs = [f for f in J.functions(srcs[file] if isinstance(srcs, dict) else srcs, name)
          if (arity is None or len(f["pos"]) == arity) and (typ is None or (f["pos"] and re.search(typ, f["pos"][0])))]
    assert fs, (file, name, arity, typ)
    return fs[0]["line"]


rows = []
R = lambda f, n, a=None, t=None: f"`src/{f}:{jl(ref, f, n, a, t)}`"
G = lambda n, a=None, t=None: f"`:{jl(glue, None, n, a, t)}`"
rows.append(("keywords + defaults of `LBFGSOperator` / `InverseLBFGSOperator` (`mem = 5, scaling = true, damped = false, σ₂ = 0.99, σ₃ = 10.0`; `inverse` accepted, ignored)",
             R("lbfgs.jl", "LBFGSData", 2), G("mxqn") + " `mxqn`", f"`qn.py:{pyline(qn.LBFGSOperator)}`, `:{pyline(qn.InverseLBFGSOperator)}`, `_kw` `:{pyline(qn._kw)}`"))
rows.append(("keywords + defaults of `LSR1Operator` (`mem = 5, scaling = true` — the code, not the docstring)",
             R("lsr1.jl", "LSR1Data", 2), G("lsr1") + " `lsr1`", f"`qn.py:{pyline(qn.LSR1Operator)}`"))
for a, what in ((3, "`push!(op, s, y)`: damped → `push!(op, s, y, similar(s))`"), (4, "`push!(op, s, y, Bs)`: undamped / inverse → `error`"),
                (6, "`push!(op, s, y, α, g, Bs)`: undamped / forward → `error`"), (5, "`push!(op, s, y, α, g)` → `push!(…, similar(g))`")):
    pat = {3: r"len\(args\) == 0", 4: r"len\(args\) == 1", 5: r"len\(args\) in \(2, 3\)", 6: r"len\(args\) in \(2, 3\)"}[a]
    rows.append((what, R("lbfgs.jl", "push!", a, "LBFGSOperator"), G("push!", a, "MXQNOperator"), f"`qn.py:{pyline(qn.push, pat)}`"))
rows.append(("`push!(op::LSR1Operator, s, y)` is the only L-SR1 method (other arities: `MethodError`)", R("lsr1.jl", "push!", 3, "LSR1Operator"),
             G("push!", 4, "MXQNOperator") + " (`op.kind == 2 && throw(MethodError…)`)", f"`qn.py:{pyline(qn.push, 'LSR1Operator, s, y. takes no')}` (TypeError)"))
rows.append(("L-BFGS flags `(true, true, prod!, prod!, prod!)`; L-SR1 `(true, true, prod!, nothing, nothing)`",
             f"`src/lbfgs.jl:157,205`, `src/lsr1.jl:110`", G("mxqn") + " (`t = kind == 2 ? nothing : prod!`)", f"`qn.py:{pyline(qn._QNOperator.__init__, 'self.symmetric = self.hermitian = True')}`, `:{pyline(qn._QNOperator.__init__, 'self.tprod = self.ctprod = None')}`"))
rows.append(("`opHouseholder`: `(isreal(h), true, prod!, nothing, prod!)`", R("linalg.jl", "opHouseholder", 1), G("opHouseholder", 1), f"`leaves.py:{pyline(leaves.opHouseholder, 'LinearOperator.h.dtype')}`"))
rows.append(("`opHermitian(d, A)`: `(isreal(A), true, prod!, nothing, nothing)`, `LinearOperatorException(\"shape mismatch\")`", R("linalg.jl", "opHermitian", 2), G("opHermitian", 2),
             f"`leaves.py:{pyline(leaves.opHermitian, 'LinearOperator.U, m, m, True, True')}`, `:{pyline(leaves.opHermitian, 'LinearOperator.U, m, m, False, True')}`"))
rows.append(("`opDiagonal(d)`: `(true, isreal(d), prod!, prod!, ctprod!)`", R("special-operators.jl", "opDiagonal", 1), G("opDiagonal", 1) + " (complex `d`; real `d`: the reference's own constructor)",
             f"`leaves.py:{pyline(leaves.opDiagonal, 'True, False, prod, prod, ctprod')}`, `:{pyline(leaves.opDiagonal, 'True, True, prod, prod, prod')}`"))
rows.append(("`solve_shifted_system!`: `σ < 0` → `ArgumentError`", "`src/utilities.jl:213-215`", G("check") + " (`st == 6`: `MXLO_EDOMAIN`)", f"`qn.py:{pyline(qn.solve_shifted_system, 'nonnegative')}` (ValueError)"))
HEAD = "| what a caller observes | reference | glue `julia/LinearOperatorsMXLOExt.jl` | mirror `linearoperators.jl_amd/` |"
table = "\n".join([HEAD, "|---|---|---|---|"] + ["| " + " | ".join(r) + " |" for r in rows]) + "\n"
if "--update" in sys.argv[1:]:                      # rewrite the table inside INTEGRATION.md in place
    doc = (ROOT / "INTEGRATION.md").read_text()
    a = doc.index(HEAD)
    b = a
    for ln in doc[a:].splitlines(keepends=True):
        if not ln.startswith("|"):
            break
        b += len(ln)
    (ROOT / "INTEGRATION.md").write_text(doc[:a] + table + doc[b:])
    print("INTEGRATION.md §3a updated")
else:
    print(table, end="")
