"""The stated tolerances of DESIGN.md §2 that more than one test file uses (relative L2 norms).

Float32 quasi-Newton operators: ONE number for every apply path (forward compact / Gram / reference order, inverse
two-pass / reference order, L-SR1, diag!, the fused shifted apply). The recurrences behind an apply amplify
eps(Float32) = 6e-8 by the conditioning of the stored pairs; on the seeded well-conditioned pairs of the parity tests
the observed error is 1e-6 .. 2e-4, and over 4010 random operation sequences (tests/test_gpu_qn_fuzz.py) 4009 stay
below QN_F32 — the one that does not is pinned by name there, together with the evidence that the Float32 REFERENCE
itself is that far from the exact operator on that sequence."""
QN_F32 = 2e-3            # mul! / diag! of LBFGSOperator, InverseLBFGSOperator, LSR1Operator on Float32 data
QN_F32_SOLVE = 2e-2      # solve_shifted_system! in coefficient space vs the oracle's recursion (src/utilities.jl:207-248)
QN_F32_ROUNDTRIP = 5e-2  # B * solve(B, b) ≈ b, the reference's own kind of check (test/test_solve_shifted_system.jl)
