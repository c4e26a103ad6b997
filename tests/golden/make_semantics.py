#!/usr/bin/env python
"""Writes tests/golden/reference_semantics.json: what /root/reference/src SAYS about the quasi-Newton constructors' keywords and
defaults, the `push!` methods and what each (damped, inverse, arity) call ends in, the (symmetric, hermitian, tprod!, ctprod!)
pattern of every hot-path constructor, and the exception type of the refusals — extracted with tests/jl_semantics.py (no Julia
runtime). Build container only (the reference does not travel); the JSON is the pivot the three-way comparison
reference == Julia glue == Python mirror hangs on: tests/test_julia_semantics.py re-derives it from the reference whenever the
reference is present (the committed file must be current), compares the glue's text with it everywhere, and the GPU suite
compares the mirror's live objects and calls with it.

    python tests/golden/make_semantics.py            # rewrites the JSON
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import jl_semantics as J  # noqa: E402

if __name__ == "__main__":
    facts = J.reference_facts("/root/reference/src")
    facts["_generated_by"] = "tests/golden/make_semantics.py from LinearOperators.jl src/ (lbfgs.jl, lsr1.jl, special-operators.jl, linalg.jl, cat.jl, utilities.jl, operations.jl)"
    with open(os.path.join(HERE, "reference_semantics.json"), "w") as f:
        json.dump(facts, f, indent=1, ensure_ascii=False, sort_keys=True)
        f.write("\n")
    print("wrote", os.path.join(HERE, "reference_semantics.json"))
