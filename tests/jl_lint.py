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
