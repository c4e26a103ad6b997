"""Semantic facts out of Julia sources WITHOUT a Julia runtime (helper of tests/test_julia_semantics.py and
tests/golden/make_semantics.py; not a test module).

Julia is not in this image, so neither the reference (/root/reference/src, build container only) nor the reference-side
glue (julia/LinearOperatorsMXLOExt.jl) can be executed. What can be done is to read both texts with the same small parser
and compare what they SAY about the things a caller observes:

* keyword names and default values of the quasi-Newton constructors,
* the set of `push!` methods per operator type and, for every (damped, inverse, arity), what a call ends in — a plain
  update, Powell damping in forward or inverse form, an ErrorException, a MethodError — by evaluating the guards
  (`if !op.data.damped error(...) elseif op.inverse error(...)`, `return push!(op, s, y, similar(s))`, ...) of each method
  body over all cases,
* the (symmetric, hermitian, tprod!, ctprod!) pattern every constructor hands to `LinearOperator{T,S}(...)`.

The parser understands exactly the subset those functions are written in (block and one-line function definitions,
`where` clauses, keyword lists with defaults, if / elseif / else / end, `cond && throw(...)`); anything else raises, so a
reference that changes shape fails loudly instead of being half-read.
"""
from __future__ import annotations

import re

OPEN = re.compile(r"\b(function|if|for|while|begin|let|do|try|struct|quote|macro|module)\b")


def strip_comments(src: str) -> str:
    out = []
    in_doc = False
    for line in src.splitlines():
        if line.strip().startswith('"""'):
            if line.strip().count('"""') == 1:
                in_doc = not in_doc
            out.append("")
            continue
        if in_doc:
            out.append("")
            continue
        in_str, k = False, 0
        while k < len(line):
            c = line[k]
            if c == '"' and (k == 0 or line[k - 1] != "\\"):
                in_str = not in_str
            elif c == "#" and not in_str:
                break
            k += 1
        out.append(line[:k].rstrip())
    return "\n".join(out)


def split_top(s: str, sep: str = ",") -> list[str]:
    out, depth, cur, in_str = [], 0, [], False
    for i, ch in enumerate(s):
        if ch == '"' and (i == 0 or s[i - 1] != "\\"):
            in_str = not in_str
        if not in_str:
            if ch in "([{":
                depth += 1
            elif ch in ")]}":
                depth -= 1
        if ch == sep and depth == 0 and not in_str:
            out.append("".join(cur).strip())
            cur = []
        else:
            cur.append(ch)
    last = "".join(cur).strip()
    if last:
        out.append(last)
    return out


def _balanced(src: str, i: int) -> int:
    """index just past the ')' matching the '(' at src[i]"""
    assert src[i] == "("
    depth, in_str = 0, False
    while True:
        ch = src[i]
        if ch == '"' and src[i - 1] != "\\":
            in_str = not in_str
        if not in_str:
            depth += ch in "([{"
            depth -= ch in ")]}"
        i += 1
        if depth == 0:
            return i


def _block_end(src: str, i: int) -> int:
    """index just past the `end` closing the block whose body starts at src[i] (depth 1)"""
    depth, brackets = 1, 0
    tok = re.compile(r"\b(function|if|for|while|begin|let|do|try|struct|quote|macro|module|end)\b|[\[\]]|\"(?:[^\"\\]|\\.)*\"")
    for m in tok.finditer(src, i):
        t = m.group(0)
        if t == "[":
            brackets += 1
        elif t == "]":
            brackets -= 1
        elif t.startswith('"'):
            continue
        elif t == "end":
            if brackets == 0:                      # a[end] is an index, not a block end
                depth -= 1
                if depth == 0:
                    return m.end()
        elif brackets == 0:
            # `if` / `for` used as a generator / ternary inside brackets never opens a block at bracket depth 0 here
            depth += 1
    raise AssertionError("unterminated block")


def functions(src: str, name: str) -> list[dict]:
    """Every definition of `name` in `src` (comments already stripped): block form and one-line form."""
    out = []
    pat = re.compile(rf"(?m)^(?P<indent>[ \t]*)(?:@\w+\s+)?(?P<kw>function\s+)?(?:[A-Za-z_.]+\.)?{re.escape(name)}(?P<tp>\{{[^}}\n]*\}})?\(")
    for m in pat.finditer(src):
        lp = m.end() - 1
        rp = _balanced(src, lp)
        args = src[lp + 1: rp - 1]
        rest = src[rp:]
        wm = re.match(r"\s*where\s*(\{(?:[^{}]|\{[^{}]*\})*\}|\w+(\s*<:\s*[\w{}, .]+)?)", rest)
        where = wm.group(0).strip() if wm else ""
        after = rp + (wm.end() if wm else 0)
        if m.group("kw"):
            end = _block_end(src, after)
            body = src[after: end - 3]
        else:
            eq = re.match(r"\s*=(?!=)", src[after:])
            if not eq:
                continue                           # a call at the start of a line, not a definition
            start = after + eq.end()
            # the expression: to the end of the line, continued while brackets are open or the next line is more indented
            i, depth, in_str = start, 0, False
            while i < len(src):
                ch = src[i]
                if ch == '"' and src[i - 1] != "\\":
                    in_str = not in_str
                if not in_str:
                    depth += ch in "([{"
                    depth -= ch in ")]}"
                if ch == "\n" and depth <= 0:
                    nxt = src[i + 1: src.find("\n", i + 1) if src.find("\n", i + 1) >= 0 else len(src)]
                    ind = len(nxt) - len(nxt.lstrip())
                    if not nxt.strip() or ind <= len(m.group("indent")):
                        break
                i += 1
            body = src[start:i]
        parts = split_top(args, ";")
        pos = split_top(parts[0]) if parts and args.strip() and not args.strip().startswith(";") else []
        if args.strip().startswith(";"):
            pos, parts = [], ["", args.strip()[1:]]
        kws = {}
        if len(parts) > 1:
            for kw in split_top(parts[1]):
                if kw.endswith("..."):
                    kws[kw] = None
                    continue
                km = re.match(r"^([^\s:=]+)\s*(?:::\s*([^=]+?))?\s*(?:=\s*(.+))?$", kw, flags=re.S)
                assert km, f"keyword not understood: {kw!r}"
                kws[km.group(1)] = km.group(3).strip() if km.group(3) is not None else None
        out.append({"name": name, "pos": pos, "kw": kws, "where": where, "body": body,
                    "line": src.count("\n", 0, m.start()) + 1})
    return out


def param_name(p: str) -> str:
    return p.split("::")[0].strip()


def param_type(p: str) -> str:
    return p.split("::", 1)[1].strip() if "::" in p else ""


# ------------------------------------------------------------------------------------------------ push! decision tables
def _jl_cond(cond: str, env: dict) -> bool:
    c = cond.strip()
    c = c.replace("op.data.damped", " damped ").replace("op.inverse", " inverse ")
    c = re.sub(r"op\.kind\s*==\s*2", " lsr1 ", c)
    c = c.replace("&&", " and ").replace("||", " or ")
    c = re.sub(r"!\s*(?=[\w( ])", " not ", c)
    if not re.fullmatch(r"[\sa-z0-9()]*", c) or not set(re.findall(r"[a-z0-9]+", c)) <= {"damped", "inverse", "lsr1", "not", "and", "or"}:
        raise AssertionError(f"guard condition not understood: {cond!r}")
    return bool(eval(c, {"__builtins__": {}}, dict(env)))


def _action(lines: list[str]):
    """What a guard branch does, from its first significant line; None when it is ordinary code (not a guard)."""
    for ln in lines:
        t = ln.strip()
        if not t:
            continue
        if t.startswith("error("):
            return ("raise", "ErrorException")
        m = re.match(r"throw\((\w+)\(", t)
        if m:
            return ("raise", m.group(1))
        m = re.match(r"(?:return\s+)?push!\((.*)\)\s*$", t)
        if m:
            return ("redirect", len(split_top(m.group(1))))
        return None
    return None


def push_guards(body: str) -> tuple[list, tuple]:
    """(guards, terminal) of one push! method body: guards = [(condition text or None for `else`, action)], in order;
    terminal = what the body does when no guard fires."""
    lines = body.splitlines()
    guards, i = [], 0
    while i < len(lines):
        t = lines[i].strip()
        m = re.match(r"(.+?)\s*&&\s*(throw\(.*|error\(.*|return\s+push!\(.*)$", t)
        if m and not t.startswith("if "):
            act = _action([m.group(2)])
            if act:
                guards.append(([m.group(1)], act))
            i += 1
            continue
        if t.startswith("if "):
            depth, j = 1, i + 1
            branches = [[t[3:].strip(), []]]
            while j < len(lines) and depth:
                u = lines[j].strip()
                if depth == 1 and u.startswith("elseif "):
                    branches.append([u[7:].strip(), []])
                elif depth == 1 and u == "else":
                    branches.append([None, []])
                elif u == "end" or u.startswith("end "):
                    depth -= 1
                else:
                    if re.match(r"(if|for|while|begin|let|try)\b", u):
                        depth += 1
                    branches[-1][1].append(u)
                j += 1
            acts = [(_c, _action(_b)) for _c, _b in branches]
            if all(a is not None for _, a in acts):
                prior = []
                for cnd, a in acts:
                    # a branch fires when its own condition holds and none of the earlier ones did
                    guards.append((prior + [cnd] if cnd is not None else prior + [None], a))
                    if cnd is not None:
                        prior = prior + [f"!({cnd})"]
            i = j
            continue
        i += 1
    text = body
    red = re.search(r"(?m)^\s*push!\((.*)\)\s*$", text)
    ccall = re.search(r"ccall\(\(:(mxlo_qn_push\w*)", text)
    if ccall:
        terminal = ("update", {"mxlo_qn_push": "plain", "mxlo_qn_push_damped_fwd": "damped_fwd",
                               "mxlo_qn_push_damped_inv": "damped_inv"}[ccall.group(1)])
    elif "push_common!" in text or "data.insert" in text:
        if re.search(r"mul!\(Bs,\s*op,\s*s", text):
            terminal = ("update", "damped_fwd")
        elif re.search(r"Bs\s*\.=\s*-α", text):
            terminal = ("update", "damped_inv")
        else:
            terminal = ("update", "plain")
    elif red:
        terminal = ("redirect", len(split_top(red.group(1))))
    else:
        raise AssertionError("push! body without a recognisable end: " + body[:200])
    return guards, terminal


def push_methods(src: str, op_type_pattern: str) -> dict[int, dict]:
    """arity -> {guards, terminal} for the push! methods whose first parameter's type matches `op_type_pattern`."""
    out = {}
    for f in functions(src, "push!"):
        if not f["pos"] or not re.search(op_type_pattern, param_type(f["pos"][0])):
            continue
        g, t = push_guards(f["body"])
        assert len(f["pos"]) not in out, f"two push! methods of arity {len(f['pos'])}"
        out[len(f["pos"])] = {"guards": g, "terminal": t, "line": f["line"]}
    return out


def push_outcome(methods: dict[int, dict], arity: int, env: dict, depth: int = 0) -> str:
    if arity not in methods:
        return "MethodError"
    assert depth < 4, "push! redirects in a circle"
    for conds, act in methods[arity]["guards"]:
        fire = True
        for c in conds:
            if c is None:
                continue
            neg = c.startswith("!(") and c.endswith(")")
            v = _jl_cond(c[2:-1] if neg else c, env)
            if (not v) if neg else v:
                continue
            fire = False
            break
        if fire:
            if act[0] == "raise":
                return act[1]
            return push_outcome(methods, act[1], env, depth + 1)
    t = methods[arity]["terminal"]
    if t[0] == "redirect":
        return push_outcome(methods, t[1], env, depth + 1)
    return t[1]


PUSH_CASES = [(kind, damped, arity) for kind in ("fwd", "inv") for damped in (False, True) for arity in (3, 4, 5, 6)] + \
             [("lsr1", False, arity) for arity in (3, 4, 5, 6)]


def push_table(lbfgs_methods: dict, lsr1_methods: dict) -> dict[str, str]:
    out = {}
    for kind, damped, arity in PUSH_CASES:
        methods = lsr1_methods if kind == "lsr1" else lbfgs_methods
        env = {"damped": damped, "inverse": kind == "inv", "lsr1": kind == "lsr1"}
        out[f"{kind}/damped={int(damped)}/arity={arity}"] = push_outcome(methods, arity, env)
    return out


# ------------------------------------------------------------------------------------------ constructor flag patterns
def _canon_flag(expr: str, env: dict) -> bool:
    e = expr.strip()
    e = re.sub(r"isreal\(\w+\)", " real ", e)
    e = re.sub(r"\bT\s*<:\s*Real\b", " real ", e)
    e = re.sub(r"\bnrow\s*==\s*ncol\b", " square ", e)
    e = e.replace("true", " True ").replace("false", " False ")
    if not set(re.findall(r"[A-Za-z]+", e)) <= {"real", "square", "True", "False"}:
        raise AssertionError(f"flag expression not understood: {expr!r}")
    return bool(eval(e, {"__builtins__": {}}, dict(env)))


def _canon_fn(expr: str, prod: str, assigns: dict, env: dict) -> str:
    e = expr.strip()
    if e != prod and re.search(r"\?\s*nothing\s*:|^nothing$", assigns.get(e, "")):
        e = assigns[e]                                        # a name bound to `nothing` or to `cond ? nothing : f`
    m = re.match(r"(.+?)\?\s*nothing\s*:\s*(.+)$", e)         # t = kind == 2 ? nothing : prod!
    if m:
        cond = re.sub(r"kind\s*==\s*2", " lsr1 ", m.group(1))
        e = "nothing" if eval(cond, {"__builtins__": {}}, dict(env)) else m.group(2).strip()
    if e == "nothing":
        return "nothing"
    return "prod" if e == prod else "own"


def constructor_patterns(body: str, ctor_regex: str, env: dict) -> set[tuple]:
    """(symmetric, hermitian, tprod, ctprod) of every `Ctor{...}(nrow, ncol, symmetric, hermitian, prod!, tprod!, ctprod!, ...)`
    call in `body`: flags evaluated under `env` (real / square / lsr1), tprod / ctprod as 'nothing' | 'prod' (the same
    object as prod!) | 'own'."""
    assigns = {m.group(1): m.group(2).strip() for m in re.finditer(r"(?m)^\s*(\w+!?)\s*=\s*([^=\n].*)$", body)}
    out = set()
    for m in re.finditer(ctor_regex, body):
        lp = body.index("(", m.end() - 1)
        args = split_top(body[lp + 1: _balanced(body, lp) - 1])
        if len(args) < 7:
            continue
        sym, herm, prod, t, ct = args[2], args[3], args[4], args[5], args[6]
        out.add((_canon_flag(sym, env), _canon_flag(herm, env), _canon_fn(t, prod.strip(), assigns, env),
                 _canon_fn(ct, prod.strip(), assigns, env)))
    return out


LINOP_CTOR = r"LinearOperator\{(?:[^{}]|\{[^{}]*\})*\}\("


def method_scenarios(f: dict) -> list[bool]:
    """element-type scenarios (real?) a glue method covers, from the bound of `T` in its where clause"""
    m = re.search(r"\bT\s*<:\s*(\w+)", f["where"])
    if m and m.group(1) == "CplxT":
        return [False]
    if m and m.group(1) == "RealT":
        return [True]
    return [True, False]


# ------------------------------------------------------------------------------------------ the facts, per source tree
ENV0 = {"real": True, "square": True, "lsr1": False}
# (key, file, function, constructor regex, positional arity or None, scenarios)
FLAG_SITES = [
    ("opEye/square", "special-operators.jl", "opEye", LINOP_CTOR, 2, [ENV0]),
    ("opEye/rectangular", "special-operators.jl", "opEye", LINOP_CTOR, 3, [dict(ENV0, square=False)]),
    ("opOnes", "special-operators.jl", "opOnes", LINOP_CTOR, 3, [ENV0, dict(ENV0, square=False)]),
    ("opZeros", "special-operators.jl", "opZeros", LINOP_CTOR, 3, [ENV0, dict(ENV0, square=False)]),
    ("opDiagonal(d)", "special-operators.jl", "opDiagonal", LINOP_CTOR, 1, [ENV0, dict(ENV0, real=False)]),
    ("opDiagonal(nrow,ncol,d)", "special-operators.jl", "opDiagonal", LINOP_CTOR, 3, [dict(ENV0, square=False), dict(ENV0, real=False, square=False)]),
    ("opRestriction", "special-operators.jl", "opRestriction", LINOP_CTOR, 2, [dict(ENV0, square=False)]),
    ("opHouseholder", "linalg.jl", "opHouseholder", LINOP_CTOR, 1, [ENV0, dict(ENV0, real=False)]),
    ("opHermitian(d,A)", "linalg.jl", "opHermitian", LINOP_CTOR, 2, [ENV0, dict(ENV0, real=False)]),
    ("InverseLBFGSOperator", "lbfgs.jl", "InverseLBFGSOperator", r"LBFGSOperator\{T\}\(", 2, [ENV0]),
    ("LBFGSOperator", "lbfgs.jl", "LBFGSOperator", r"LBFGSOperator\{T\}\(", 2, [ENV0]),
    ("LSR1Operator", "lsr1.jl", "LSR1Operator", r"LSR1Operator\{T\}\(", 2, [dict(ENV0, lsr1=True)]),
    ("hcat", "cat.jl", "hcat", LINOP_CTOR, 2, [dict(ENV0, square=False)]),
    ("vcat", "cat.jl", "vcat", LINOP_CTOR, 2, [dict(ENV0, square=False)]),
]


def env_key(env: dict) -> str:
    return ("real" if env["real"] else "complex") + ("" if env["square"] else ",rectangular")


def flag_row(pattern: tuple) -> dict:
    sym, herm, t, ct = pattern
    return {"symmetric": sym, "hermitian": herm, "tprod": "nothing" if t == "nothing" else "set",
            "ctprod": "nothing" if ct == "nothing" else "set"}


def reference_facts(src_dir) -> dict:
    """Everything tests/golden/reference_semantics.json holds, read out of a LinearOperators.jl `src/` directory."""
    import pathlib
    src = {p.name: strip_comments(p.read_text()) for p in pathlib.Path(src_dir).glob("*.jl")}
    facts = {"qn_keywords": {}, "push": {}, "flags": {}, "errors": {}}
    for data, file in (("LBFGSData", "lbfgs.jl"), ("LSR1Data", "lsr1.jl")):
        full = [f for f in functions(src[file], data) if len(f["pos"]) == 2]
        assert len(full) == 1, f"{data}(T, n; ...) not found once"
        facts["qn_keywords"][data] = full[0]["kw"]
    lb, ls = push_methods(src["lbfgs.jl"], r"LBFGSOperator"), push_methods(src["lsr1.jl"], r"LSR1Operator")
    facts["push"]["arities"] = {"LBFGSOperator": sorted(lb), "LSR1Operator": sorted(ls)}
    facts["push"]["outcomes"] = push_table(lb, ls)
    for key, file, fname, ctor, npos, envs in FLAG_SITES:
        fs = [f for f in functions(src[file], fname) if len(f["pos"]) == npos and "::Colon" not in f["pos"][0]]
        for env in envs:
            got = set()
            for f in fs:
                got |= constructor_patterns(f["body"], ctor, env)
            rows = {tuple(sorted(flag_row(p).items())) for p in got}
            assert len(rows) == 1, f"{key} [{env_key(env)}]: {len(rows)} distinct flag patterns in the reference: {got}"
            facts["flags"][f"{key} [{env_key(env)}]"] = dict(rows.pop())
    # refusals: which exception type guards what
    u = src["utilities.jl"]
    m = re.search(r"σ\s*<\s*0\s*&&\s*throw\((\w+)\(|if\s+σ\s*<\s*0[^\n]*\n\s*throw\((\w+)\(", u)
    facts["errors"]["solve_shifted_system!: σ < 0"] = (m.group(1) or m.group(2)) if m else None
    m = re.search(r"throw\((\w+)\(\"shape mismatch\"\)\)", src["operations.jl"])
    facts["errors"]["mul!: shape mismatch"] = m.group(1) if m else None
    m = re.search(r"throw\((\w+)\(\"shape mismatch\"\)\)", src["linalg.jl"])
    facts["errors"]["opHermitian: shape mismatch"] = m.group(1) if m else None
    return facts
