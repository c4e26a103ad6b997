"""-m gpu: randomized OPERATION SEQUENCES on the quasi-Newton operators against the oracle — the state machine behind
the handle (circular buffer wrap, rejected pairs, partially filled memory, reset!, evaluation-mode and push-mode switches,
lazy a_k panel, cached shifted-solve Gram) must never change what the operator IS. After every step the operator is
applied and compared with the oracle driven through the same sequence (fp64: 1e-8 relative to ||x||·||B||-scale)."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    nb = np.linalg.norm(b)
    return np.linalg.norm(a - b) / (nb if nb else 1.0)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("MXLO_QNFUZZ_SEEDS", "60"))))
def test_random_operation_sequences(lo, dev, seed):
    run_sequence(lo, dev, seed, torch.float64, 1e-8, 1e-7)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("MXLO_QNFUZZ32_SEEDS", "30"))))
def test_random_operation_sequences_fp32(lo, dev, seed):
    """The same state machine on Float32 data (Float64 scalars from the caller: Julia's mixed-precision rule)."""
    run_sequence(lo, dev, seed, torch.float32, 2e-3, 2e-2)


def run_sequence(lo, dev, seed, dtype, tol, tol_solve):
    npd = np.float64 if dtype == torch.float64 else np.float32
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([1, 2, 3, 17, 130, 1025, 4099, 20_001]))
    mem = int(rng.integers(1, 9))
    kind = ["fwd", "inv", "lsr1"][seed % 3]
    if kind == "lsr1" and n < 17:
        n = 17       # SR1 with more stored pairs than dimensions is exact after n of them: every further term is
                     # a_k = y_k - B s_k = 0 up to rounding, and a_k's_k (rounding noise in the reference, exactly 0 in the
                     # Gram form) is divided by — garbage either way, so keep n > mem for L-SR1
    scaling = bool(rng.integers(2))
    if kind == "fwd":
        op, O = lo.LBFGSOperator(dtype, n, mem=mem, scaling=scaling, device=dev), oracle.LBFGS(n, mem=mem, scaling=scaling, inverse=False, dtype=npd)
    elif kind == "inv":
        op, O = lo.InverseLBFGSOperator(dtype, n, mem=mem, scaling=scaling, device=dev), oracle.LBFGS(n, mem=mem, scaling=scaling, inverse=True, dtype=npd)
    else:
        op, O = lo.LSR1Operator(dtype, n, mem=mem, scaling=scaling, device=dev), oracle.LSR1(n, mem=mem, scaling=scaling, dtype=npd)
    Dg = rng.uniform(0.5, 2.0, n)
    fl = oracle.SCALARS_F64 if dtype == torch.float32 else 0

    def check(tag):
        x, r0 = rng.uniform(-1, 1, n).astype(npd), rng.uniform(-1, 1, n).astype(npd)
        a, b = (1.0, 0.0) if rng.integers(2) else (float(rng.uniform(-2, 2)), float(rng.uniform(-2, 2)))
        res = T(r0.copy(), dev)
        lo.mul(res, op, T(x, dev), a, b)
        want = O.mul(r0.copy(), x, a, b, flags=fl).astype(np.float64)
        scale = np.linalg.norm(want) + abs(a) * np.linalg.norm(O.mul(np.empty(n, npd), x).astype(np.float64)) + abs(b) * np.linalg.norm(r0) + 1e-300
        assert np.linalg.norm(res.cpu().numpy().astype(np.float64) - want) <= tol * scale, (tag, kind, n, mem)
        assert op.data.insert == O.insert

    check("fresh")
    for step in range(int(rng.integers(5, 30))):
        c = rng.integers(12)
        if c <= 5:                                            # push a (mostly) well-conditioned pair
            s = rng.uniform(-1, 1, n).astype(npd)
            y = (Dg * s + 1e-2 * rng.standard_normal(n)).astype(npd)
            if c == 5:
                y = -y if rng.integers(2) else np.zeros(n, npd)    # negative / zero curvature: rejected by L-BFGS
            lo.push(op, T(s, dev), T(y, dev))
            O.push(s, y)
        elif c == 6:
            lo.reset(op)
            O.reset()
        elif c == 7 and kind == "inv":
            op.set_mode("reforder" if rng.integers(2) else "twopass")
        elif c == 7:
            op.set_push_mode(["gram", "reforder", "compact"][rng.integers(3)])
        elif c == 8 and kind != "inv":
            got, want = lo.diag(op).cpu().numpy().astype(np.float64), O.diag().astype(np.float64)
            assert np.linalg.norm(got - want) <= tol * (np.linalg.norm(want) + 1e-300), ("diag", kind, n, mem)
        elif c == 9 and kind == "fwd":
            bvec, sig = rng.uniform(-1, 1, n).astype(npd), npd(rng.uniform(0, 2))
            got = lo.solve_shifted_system(torch.zeros(n, dtype=dtype, device=dev), op, T(bvec, dev), sig).cpu().numpy().astype(np.float64)
            want = O.solve_shifted(np.zeros(n, npd), bvec, sig).astype(np.float64)
            assert np.linalg.norm(got - want) <= tol_solve * (np.linalg.norm(want) + 1e-300), ("solve_shifted", n, mem)
        elif c == 10:
            sig = float(rng.uniform(-1, 1))
            x = rng.uniform(-1, 1, n).astype(npd)
            got = (lo.ShiftedOperator(op, sig) * T(x, dev)).cpu().numpy().astype(np.float64)
            Bx = O.mul(np.empty(n, npd), x).astype(np.float64)
            want = Bx + sig * x.astype(np.float64)
            assert np.linalg.norm(got - want) <= tol * (np.linalg.norm(Bx) + abs(sig) * np.linalg.norm(x) + 1e-300)
        check(f"step {step} op {c}")
