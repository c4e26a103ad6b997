"""A structural checker for Julia source text (no Julia in the image: julia/*.jl has never been parsed by Julia itself).

It does NOT parse Julia. It tokenises just enough — line and nested block comments, string / triple-quoted string literals with
`$( … )` interpolation, character literals vs the adjoint operator, brackets — to check the two things a file that never ran is
most likely to get wrong: that every block keyword (`function`, `if`, `for`, `while`, `let`, `begin`, `try`, `do`, `struct`,
`module`, `quote`, `macro`, `abstract type` / `primitive type`) has its `end`, and that ( [ { close in order. Rules it relies on:
`end` / `begin` inside square brackets are indexing keywords; `for` / `if` directly inside a bracket are generator / comprehension
clauses, not blocks; `:end`, `.end` and the like are symbols / fields.

Self-test (tests/test_julia_lint.py): every file of the reference (`/root/reference/src`, `ext`, `test`: real, running Julia) must
pass, and planted defects (a dropped `end`, an extra one, a dropped bracket, an unterminated string) must be reported.
"""
import re

OPENERS = {"function", "macro", "if", "for", "while", "let", "begin", "try", "do", "struct", "module", "baremodule", "quote"}
GENERATOR_WORDS = {"for", "if"}                 # directly inside a bracket: clause of a generator / comprehension
IDENT = re.compile(r"[A-Za-z_¡-￿][A-Za-z0-9_!¡-￿]*")
CLOSE = {")": "(", "]": "[", "}": "{"}


class JlSyntaxError(Exception):
    pass


def check(text: str, name: str = "<text>", code_brackets: list | None = None) -> int:
    """Raises JlSyntaxError with file:line on the first structural defect; returns the number of blocks seen.
    code_brackets: if a list, receives the text positions of the brackets met in CODE (not in comments / strings)."""
    n = len(text)
    line = 1
    stack = []                                  # ('(' | '[' | '{' | 'block', keyword-or-bracket, line)
    blocks = 0
    prev_sig = ""                               # last significant character outside comments / whitespace
    prev_word = ""

    def err(msg, ln=None):
        raise JlSyntaxError(f"{name}:{ln if ln is not None else line}: {msg}")

    def skip_string(i, triple):
        """i: index just past the opening quote(s); returns the index just past the closing quote(s)."""
        nonlocal line
        start = line
        while i < n:
            c = text[i]
            if c == "\\":
                if i + 1 < n and text[i + 1] == "\n":
                    line += 1
                i += 2
                continue
            if c == "\n":
                line += 1
                if not triple:
                    pass                        # Julia allows a newline inside "…"
            if c == "$" and i + 1 < n and text[i + 1] == "(":
                i = skip_parens(i + 2)
                continue
            if c == '"':
                if not triple:
                    return i + 1
                if text[i:i + 3] == '"""':
                    return i + 3
            i += 1
        err("unterminated string literal", start)

    def skip_parens(i):
        """code inside $( … ): returns the index just past the matching ')'. Strings and nested parens are honoured."""
        nonlocal line
        depth, start = 1, line
        while i < n:
            c = text[i]
            if c == "\n":
                line += 1
            elif c == '"':
                triple = text[i:i + 3] == '"""'
                i = skip_string(i + (3 if triple else 1), triple)
                continue
            elif c == "(":
                depth += 1
            elif c == ")":
                depth -= 1
                if depth == 0:
                    return i + 1
            i += 1
        err("unterminated $( … ) interpolation", start)

    i = 0
    while i < n:
        c = text[i]
        if c == "\n":
            line += 1
            i += 1
            continue
        if c in " \t\r":
            i += 1
            continue
        if c == "#":
            if text[i:i + 2] == "#=":           # nested block comment
                depth, start = 1, line
                i += 2
                while i < n and depth:
                    if text[i:i + 2] == "#=":
                        depth += 1
                        i += 2
                    elif text[i:i + 2] == "=#":
                        depth -= 1
                        i += 2
                    else:
                        if text[i] == "\n":
                            line += 1
                        i += 1
                if depth:
                    err("unterminated #= … =# comment", start)
            else:
                while i < n and text[i] != "\n":
                    i += 1
            continue
        if c == '"':
            triple = text[i:i + 3] == '"""'
            i = skip_string(i + (3 if triple else 1), triple)
            prev_sig, prev_word = '"', ""
            continue
        if c == "`":
            j = text.find("`", i + 1)
            if j < 0:
                err("unterminated `…` command literal")
            line += text.count("\n", i, j)
            i = j + 1
            prev_sig, prev_word = "`", ""
            continue
        if c == "'":
            # adjoint after an identifier, a closing bracket, another adjoint or a dot-call; a character literal otherwise
            if prev_sig and (prev_sig.isalnum() or prev_sig in "_)]}'!" or ord(prev_sig) > 0xa0) and not (prev_word in OPENERS or prev_word in ("return", "in", "isa", "where")):
                i += 1
                prev_sig = "'"
                continue
            j = i + 1
            if j < n and text[j] == "\\":
                j += 2
                while j < n and text[j] != "'" and text[j] != "\n":
                    j += 1
            else:
                j += 1
                while j < n and text[j] != "'" and text[j] != "\n" and j - i < 6:
                    j += 1
            if j >= n or text[j] != "'":
                err("unterminated character literal")
            i = j + 1
            prev_sig, prev_word = "'", ""
            continue
        if c in "([{":
            if code_brackets is not None:
                code_brackets.append(i)
            stack.append((c, c, line))
            i += 1
            prev_sig, prev_word = c, ""
            continue
        if c in ")]}":
            if code_brackets is not None:
                code_brackets.append(i)
            if not stack:
                err(f"'{c}' closes nothing")
            kind, what, ln = stack.pop()
            if kind == "block":
                err(f"'{c}' while the `{what}` block opened on line {ln} is still open")
            if kind != CLOSE[c]:
                err(f"'{c}' closes the '{kind}' opened on line {ln}")
            i += 1
            prev_sig, prev_word = c, ""
            continue
        m = IDENT.match(text, i)
        if m:
            w = m.group(0)
            j = m.end()
            before = text[i - 1] if i else ""
            before2 = text[i - 2] if i > 1 else ""
            is_symbol = before == ":" and before2 != ":" and (before2 == "" or not (before2.isalnum() or before2 in "_)]}"))
            is_field = before == "."
            is_macro_name = before == "@"
            if w in ("end", "begin") and not is_symbol and not is_field and not is_macro_name:
                # indexing keyword inside [...] (no block opened inside that bracket)?
                indexing = False
                for kind, _, _ in reversed(stack):
                    if kind == "block":
                        break
                    if kind == "[":
                        indexing = True
                        break
                if indexing or (w == "end" and before == ":" and any(k == "[" for k, _, _ in stack)):
                    i = j
                    prev_sig, prev_word = w[-1], w
                    continue
            if w == "end" and not is_symbol and not is_field and not is_macro_name:
                if not stack:
                    err("`end` closes nothing")
                kind, what, ln = stack.pop()
                if kind != "block":
                    err(f"`end` inside the '{kind}' opened on line {ln}")
            elif w in OPENERS and not is_symbol and not is_field and not is_macro_name:
                top = stack[-1][0] if stack else ""
                if w in GENERATOR_WORDS and top in ("(", "[", "{"):
                    pass                        # generator / comprehension clause
                elif w == "struct" and prev_word == "mutable":
                    stack.append(("block", "mutable struct", line))
                    blocks += 1
                else:
                    stack.append(("block", w, line))
                    blocks += 1
            elif w == "type" and prev_word in ("abstract", "primitive") and not is_symbol:
                stack.append(("block", prev_word + " type", line))
                blocks += 1
            i = j
            prev_sig, prev_word = w[-1], w
            continue
        i += 1
        prev_sig = c
        if not c.isspace():
            prev_word = ""
    if stack:
        kind, what, ln = stack[-1]
        raise JlSyntaxError(f"{name}:{ln}: {'`' + what + '` block' if kind == 'block' else repr(kind)} is never closed")
    return blocks


# ---------------------------------------------------------------------------------------------------------------------------------
# Arity of calls to the file's OWN helpers (names neither the reference nor Base knows): a call whose number of positional
# arguments no definition (long form, short form, or a struct's default constructor) accepts is a MethodError waiting for the
# first run. Keyword arguments (`k = v`, or anything after `;`) are ignored, splatted calls are skipped, `do` adds one.
def strip_comments_and_strings(t: str) -> str:
    t = re.sub(r'"""[\s\S]*?"""', '""', t)
    t = re.sub(r'"(?:\\.|[^"\\])*"', '""', t)
    t = re.sub(r"#=.*?=#", "", t, flags=re.S)
    return re.sub(r"#[^\n]*", "", t)


def _match_paren(t, i):
    d = 0
    for j in range(i, len(t)):
        c = t[j]
        if c in "([{":
            d += 1
        elif c in ")]}":
            d -= 1
            if d == 0:
                return j
    return -1


def _split_top(s):
    out, d, cur, semi = [], 0, "", None
    for c in s:
        if c in "([{":
            d += 1
        elif c in ")]}":
            d -= 1
        if d == 0 and c in ",;":
            out.append(cur)
            cur = ""
            if c == ";" and semi is None:
                semi = len(out)
            continue
        cur += c
    if cur.strip():
        out.append(cur)
    return out, semi


def internal_call_arity(text: str, known_elsewhere: set) -> list:
    """[(name, positional args at the call, [(min, max) per definition], line)] for calls no definition accepts."""
    g = strip_comments_and_strings(text)
    has_default = lambda p: "=" in re.sub(r"[<>=!]=|<:|=>", "", p)
    defs, spans = {}, []
    for m in re.finditer(r"(?m)^[ \t]*(?:@inline[ \t]+)?(function[ \t]+)?((?:[A-Za-z_][A-Za-z0-9_]*\.)*)([A-Za-z_][A-Za-z0-9_!]*)[ \t]*(\{[^}\n]*\})?\(", g):
        i = m.end() - 1
        j = _match_paren(g, i)
        if j < 0:
            continue
        short = re.match(r"\s*(?:::[^=\n]+?)?\s*(?:where\s*(?:\{[^}]*\}|[A-Za-z_]\w*(?:<:[^=\s]+)?))?\s*=(?!=)", g[j + 1:j + 200])
        if not (m.group(1) or short):
            continue
        params, semi = _split_top(g[i + 1:j])
        pos = [p for p in (params[:semi] if semi is not None else params) if p.strip()]
        req = sum(1 for p in pos if not has_default(p) and not p.strip().endswith("..."))
        opt = sum(1 for p in pos if has_default(p))
        var = any(p.strip().endswith("...") for p in pos)
        defs.setdefault(m.group(3), []).append((req, 99 if var else req + opt, m.group(2)))
        spans.append((m.start(), j))
    for m in re.finditer(r"(?m)^[ \t]*(?:mutable[ \t]+)?struct[ \t]+([A-Za-z_]\w*)[^\n]*\n([\s\S]*?)^end", g):
        body = m.group(2)
        if re.search(r"\bfunction\b|\bnew\b", body):
            continue                                # inner constructors: their own definitions count
        nf = sum(1 for ln in body.split("\n") if ln.strip() and re.match(r"\s*(?:const\s+)?[A-Za-z_]\w*!?\s*(::|$)", ln))
        defs.setdefault(m.group(1), []).append((nf, nf, ""))
    internal = {n for n, v in defs.items() if n not in known_elsewhere and all(q == "" for _, _, q in v)}
    bad = []
    for m in re.finditer(r"\b([A-Za-z_][A-Za-z0-9_!]*)[ \t]*(\{[^}\n]*\})?\(", g):
        name = m.group(1)
        if name not in internal:
            continue
        if any(a <= m.start() <= b and g[a:m.start()].strip() in ("", "function", "@inline", "@inline function") for a, b in spans):
            continue                                # the definition itself
        i = m.end() - 1
        j = _match_paren(g, i)
        args, semi = _split_top(g[i + 1:j])
        pos = [a for a in (args[:semi] if semi is not None else args) if a.strip() and not re.match(r"^\s*[A-Za-z_]\w*\s*=(?!=)", a)]
        if any(a.strip().endswith("...") for a in pos):
            continue
        n = len(pos) + (1 if re.match(r"\s*do\b", g[j + 1:j + 6]) else 0)
        if not any(lo <= n <= hi for lo, hi, _ in defs[name]):
            bad.append((name, n, [(lo, hi) for lo, hi, _ in defs[name]], g.count("\n", 0, m.start()) + 1))
    return bad
