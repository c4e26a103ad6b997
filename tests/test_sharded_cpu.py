"""CPU, world_size 2, gloo: the row-sharded path.

On the GPU the per-shard arithmetic is libmxlo.so and the exchange is the all-reduce hook
(`mxlo_ctx_set_allreduce` -> RCCL). Here the SAME hook body (`sharded.make_allreduce_hook`) and the
SAME shard plan run under gloo, with the oracle standing in for the per-shard kernels (tests may use
the oracle; the product never does). What is verified: (1) the hook all-reduces `count` doubles in
place through a raw pointer and returns 0, (2) partial dots summed across row ranges + replicated
coefficient math + local combine reproduce the unsharded oracle result for Householder, forward
L-BFGS and the inverse two-loop (reference-ordered: 2m chained 1-double all-reduces), (3) every rank
ends with bit-identical scalars."""
import os
import subprocess
import sys
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys, ctypes
    import numpy as np, torch, torch.distributed as dist
    sys.path.insert(0, %(root)r)
    import __graft_entry__ as g
    lo = g.load_package()
    import oracle
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    hook = lo.sharded.make_allreduce_hook(None, cuda=False)

    def allreduce_inplace(arr):
        assert arr.dtype == np.float64 and arr.flags["C_CONTIGUOUS"]
        rc = hook(None, arr.ctypes.data, arr.size, None)
        assert rc == 0
        return arr

    n, mem = 10007, 6
    rng = np.random.default_rng(123)            # same stream on every rank: replicated global data
    plan = lo.sharded.ShardPlan(n, world)
    a, b = plan.lo(rank), plan.hi(rank)
    h = rng.standard_normal(n); h /= np.linalg.norm(h)
    v = rng.uniform(-1, 1, n); r0 = rng.uniform(-1, 1, n)

    # ---- Householder: partial h'v -> all-reduce(1 double) -> local update
    part = np.array([float(oracle.dot(np.ascontiguousarray(h[a:b]), np.ascontiguousarray(v[a:b])))])
    allreduce_inplace(part)
    c = 2.0 * part[0]
    loc = 2.0 * (v[a:b] - c * h[a:b]) + (-3.0) * r0[a:b]
    full = oracle.householder_mul(r0.copy(), h, v, 2.0, -3.0)
    err_h = np.linalg.norm(loc - full[a:b]) / np.linalg.norm(full[a:b])

    # ---- forward L-BFGS: state built unsharded (replicated), apply sharded
    Bo = oracle.LBFGS(n, mem=mem, scaling=True, inverse=False)
    Ho = oracle.LBFGS(n, mem=mem, scaling=True, inverse=True)
    for _ in range(mem + 2):
        s = rng.uniform(-1, 1, n); y = s * rng.uniform(0.5, 2.0, n) + 1e-2 * rng.standard_normal(n)
        Bo.push(s, y); Ho.push(s, y)
    x = rng.uniform(-1, 1, n)
    order = [(Bo.insert - 1 + i) %% mem for i in range(mem) if Bo.ys[(Bo.insert - 1 + i) %% mem] != 0]
    dots = np.zeros(2 * len(order))
    for i, k in enumerate(order):
        dots[2 * i] = oracle.dot(np.ascontiguousarray(Bo.b[k, a:b]), np.ascontiguousarray(x[a:b]))
        dots[2 * i + 1] = oracle.dot(np.ascontiguousarray(Bo.a[k, a:b]), np.ascontiguousarray(x[a:b]))
    allreduce_inplace(dots)                      # ONE all-reduce of 2m doubles per apply
    q = x[a:b] / Bo.scaling_factor
    for i, k in enumerate(order):
        q = q + ((dots[2 * i] * Bo.b[k, a:b]) - (dots[2 * i + 1] * Bo.a[k, a:b]))
    full = Bo.mul(np.empty(n), x)
    err_f = np.linalg.norm(q - full[a:b]) / np.linalg.norm(full[a:b])

    # ---- inverse two-loop, reference order: 2m chained 1-double all-reduces
    q = x[a:b].copy(); al = {}
    newest_first = [(Ho.insert - 2 - i) %% mem for i in range(mem)]
    for k in newest_first:
        if Ho.ys[k] != 0:
            d = np.array([float(oracle.dot(np.ascontiguousarray(Ho.s[k, a:b]), np.ascontiguousarray(q)))])
            allreduce_inplace(d)
            al[k] = d[0] / Ho.ys[k]
            q = q - al[k] * Ho.y[k, a:b]
    q = q * Ho.scaling_factor
    for k in reversed(newest_first):
        if Ho.ys[k] != 0:
            d = np.array([float(oracle.dot(np.ascontiguousarray(Ho.y[k, a:b]), np.ascontiguousarray(q)))])
            allreduce_inplace(d)
            q = q + (al[k] - d[0] / Ho.ys[k]) * Ho.s[k, a:b]
    full = Ho.mul(np.empty(n), x)
    err_i = np.linalg.norm(q - full[a:b]) / np.linalg.norm(full[a:b])

    # ---- scalars bit-identical on every rank
    t = torch.tensor([part[0], dots.sum()], dtype=torch.float64)
    gathered = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(gathered, t)
    same = all(torch.equal(gathered[0], gt) for gt in gathered)
    print("RESULT", rank, err_h, err_f, err_i, int(same), flush=True)
    dist.destroy_process_group()
''')


def test_row_sharded_path_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29653", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
    for o in outs:
        line = [l for l in o.splitlines() if l.startswith("RESULT")][0].split()
        err_h, err_f, err_i, same = float(line[2]), float(line[3]), float(line[4]), int(line[5])
        assert err_h <= 1e-12 and err_f <= 1e-12 and err_i <= 1e-10 and same == 1, o
