"""-m gpu: randomized OPERATION SEQUENCES on the quasi-Newton operators against the oracle — the state machine behind
the handle (circular buffer wrap, rejected pairs, partially filled memory, reset!, evaluation-mode and push-mode switches,
lazy a_k panel, cached shifted-solve Gram) must never change what the operator IS. After every step the operator is
applied and compared with the oracle driven through the same sequence (fp64: 1e-8 relative to ||x||·||B||-scale)."""
import numpy as np
import pytest
import torch

import oracle
from tolerances import QN_F32_FUZZ_LBFGS, QN_F32_FUZZ_LSR1, QN_F32_SOLVE, observe

QN_F32 = QN_F32_FUZZ_LSR1      # the pinned ill-conditioned sequence is an L-SR1 one

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
    if seed == ILL_CONDITIONED_F32_SEED:
        pytest.skip("pinned by name below: test_fp32_lsr1_ill_conditioned_sequence")
    run_sequence(lo, dev, seed, torch.float32, QN_F32_FUZZ_LSR1 if seed % 3 == 2 else QN_F32_FUZZ_LBFGS, QN_F32_SOLVE)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("MXLO_QNFUZZ_PERSIST_SEEDS", "9"))))
def test_random_operation_sequences_in_the_range_of_the_persistent_apply(lo, dev, seed):
    """The same state machine at n = 2^19 ... 1.4e6, where an apply is ONE persistent launch with x and the first columns
    parked in LDS between its two phases (csrc/qn.hip: qn_apply_persist_kernel; `qn_persist_min_bytes` lowered so that
    partially filled memories take it too): chunk counts per workgroup 2 ... 6, ragged last chunks, every count of parked
    columns, fp64 and (every third seed) fp32."""
    from linearoperators_jl_amd.device import get_ctx
    ctx = get_ctx(dev)
    ctx.tune("qn_persist_min_bytes", 0)
    try:
        f32 = seed % 3 == 2
        run_sequence(lo, dev, 50_000 + seed, torch.float32 if f32 else torch.float64,
                     (QN_F32_FUZZ_LSR1 if (50_000 + seed) % 3 == 2 else QN_F32_FUZZ_LBFGS) if f32 else 1e-8, QN_F32_SOLVE if f32 else 1e-7,
                     sizes=[1 << 19, (1 << 19) + 5, 700_001, 1 << 20, (1 << 20) + 3, 1_400_007], max_steps=12)
    finally:
        ctx.tune("qn_persist_min_bytes", 32 << 20)


ILL_CONDITIONED_F32_SEED = 1154


def test_fp32_lsr1_ill_conditioned_sequence(lo, dev):
    """The one sequence out of 4010 (MXLO_QNFUZZ32_SEEDS=4010) whose Float32 apply leaves the stated tolerance: seed 1154,
    LSR1Operator n = 17, mem = 4, error 1.6e-2 of the result scale at step 17. It is a conditioning effect, not a kernel
    defect — with n = 17 and 4 stored pairs the SR1 denominators a_k's_k = (y_k - B_k s_k)'s_k nearly cancel, and Float32
    rounding of ANY evaluation order moves the operator by percents. Evidence asserted here:
      * the same sequence on Float64 data agrees with the oracle to 1e-8 (the state machine and kernels are right);
      * at every check of the Float32 run, the GPU result is no further from the Float32 oracle than 4x the distance
        between the Float32 oracle and the Float64 oracle driven by the same (Float32-representable) pairs — the
        reference restatement ITSELF is that far from the exact operator — or it is inside QN_F32;
      * that distance does exceed QN_F32 somewhere in the sequence (this really is the ill-conditioned seed)."""
    run_sequence(lo, dev, ILL_CONDITIONED_F32_SEED, torch.float64, 1e-8, 1e-7)
    rec = []
    run_sequence(lo, dev, ILL_CONDITIONED_F32_SEED, torch.float32, QN_F32, QN_F32_SOLVE, shadow=rec)
    assert rec and max(amp for _, _, amp in rec) > QN_F32, rec
    for tag, err, amp in rec:
        assert err <= max(QN_F32, 4.0 * amp), (tag, err, amp)
    worst = max(rec, key=lambda r: r[1])
    print(f"\nseed {ILL_CONDITIONED_F32_SEED}: worst GPU-vs-oracle32 {worst[1]:.2e} at {worst[0]} where oracle32-vs-oracle64 is {worst[2]:.2e}")


def run_sequence(lo, dev, seed, dtype, tol, tol_solve, shadow=None, sizes=None, max_steps=30):
    """shadow: a list -> Float32 runs also drive a Float64 oracle with the same pairs and, instead of asserting the apply
    tolerance, record (tag, |gpu - oracle32| / scale, |oracle32 - oracle64| / scale) per check."""
    npd = np.float64 if dtype == torch.float64 else np.float32
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice(sizes if sizes else [1, 2, 3, 17, 130, 1025, 4099, 20_001]))
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
    O64 = None
    if shadow is not None and dtype == torch.float32:
        O64 = (oracle.LSR1(n, mem=mem, scaling=scaling, dtype=np.float64) if kind == "lsr1" else
               oracle.LBFGS(n, mem=mem, scaling=scaling, inverse=(kind == "inv"), dtype=np.float64))
    Dg = rng.uniform(0.5, 2.0, n)
    fl = oracle.SCALARS_F64 if dtype == torch.float32 else 0

    def check(tag):
        x, r0 = rng.uniform(-1, 1, n).astype(npd), rng.uniform(-1, 1, n).astype(npd)
        a, b = (1.0, 0.0) if rng.integers(2) else (float(rng.uniform(-2, 2)), float(rng.uniform(-2, 2)))
        res = T(r0.copy(), dev)
        lo.mul(res, op, T(x, dev), a, b)
        want = O.mul(r0.copy(), x, a, b, flags=fl).astype(np.float64)
        scale = np.linalg.norm(want) + abs(a) * np.linalg.norm(O.mul(np.empty(n, npd), x).astype(np.float64)) + abs(b) * np.linalg.norm(r0) + 1e-300
        err = np.linalg.norm(res.cpu().numpy().astype(np.float64) - want)
        if O64 is not None:
            want64 = O64.mul(r0.astype(np.float64), x.astype(np.float64), a, b)
            shadow.append((tag, err / scale, np.linalg.norm(want - want64) / scale))
        else:
            observe(f"fuzz {kind} mul!", err / scale, dtype == torch.float32)
            assert err <= tol * scale, (tag, kind, n, mem)
        assert op.data.insert == O.insert

    check("fresh")
    for step in range(int(rng.integers(5, max_steps))):
        c = rng.integers(12)
        if c <= 5:                                            # push a (mostly) well-conditioned pair
            s = rng.uniform(-1, 1, n).astype(npd)
            y = (Dg * s + 1e-2 * rng.standard_normal(n)).astype(npd)
            if c == 5:
                y = -y if rng.integers(2) else np.zeros(n, npd)    # negative / zero curvature: rejected by L-BFGS
            lo.push(op, T(s, dev), T(y, dev))
            O.push(s, y)
            if O64 is not None:
                O64.push(s.astype(np.float64), y.astype(np.float64))
        elif c == 6:
            lo.reset(op)
            O.reset()
            if O64 is not None:
                O64.reset()
        elif c == 7 and kind == "inv":
            op.set_mode("reforder" if rng.integers(2) else "twopass")
        elif c == 7:
            op.set_push_mode(["gram", "reforder", "compact"][rng.integers(3)])
        elif c == 8 and kind != "inv":
            got, want = lo.diag(op).cpu().numpy().astype(np.float64), O.diag().astype(np.float64)
            if O64 is None:
                observe(f"fuzz {kind} diag!", np.linalg.norm(got - want) / (np.linalg.norm(want) + 1e-300), dtype == torch.float32)
            assert O64 is not None or np.linalg.norm(got - want) <= tol * (np.linalg.norm(want) + 1e-300), ("diag", kind, n, mem)
        elif c == 9 and kind == "fwd":
            bvec, sig = rng.uniform(-1, 1, n).astype(npd), npd(rng.uniform(0, 2))
            got = lo.solve_shifted_system(torch.zeros(n, dtype=dtype, device=dev), op, T(bvec, dev), sig).cpu().numpy().astype(np.float64)
            want = O.solve_shifted(np.zeros(n, npd), bvec, sig).astype(np.float64)
            observe("fuzz fwd solve_shifted_system!", np.linalg.norm(got - want) / (np.linalg.norm(want) + 1e-300), dtype == torch.float32 and O64 is None)
            assert np.linalg.norm(got - want) <= tol_solve * (np.linalg.norm(want) + 1e-300), ("solve_shifted", n, mem)
        elif c == 10:
            sig = float(rng.uniform(-1, 1))
            x = rng.uniform(-1, 1, n).astype(npd)
            got = (lo.ShiftedOperator(op, sig) * T(x, dev)).cpu().numpy().astype(np.float64)
            Bx = O.mul(np.empty(n, npd), x).astype(np.float64)
            want = Bx + sig * x.astype(np.float64)
            if O64 is None:
                observe(f"fuzz {kind} shifted mul!", np.linalg.norm(got - want) / (np.linalg.norm(Bx) + abs(sig) * np.linalg.norm(x) + 1e-300), dtype == torch.float32)
            assert O64 is not None or np.linalg.norm(got - want) <= tol * (np.linalg.norm(Bx) + abs(sig) * np.linalg.norm(x) + 1e-300)
        check(f"step {step} op {c}")
