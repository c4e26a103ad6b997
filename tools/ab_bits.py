#!/usr/bin/env python
"""A/B of two builds of libmxlo.so by RESULT BITS: runs a fixed, seeded set of applies under the library named by
MXLO_LIB_PATH (default: the in-tree one) and prints one sha256 per case; `ab_bits.py --compare A.so B.so` runs itself
twice and reports the cases whose bits differ. Used when a kernel change claims "same summation tree, same bits"
(round 6: opHermitian's column butterfly off ds_bpermute; the push! passes; the dense block apply).

    gpurun -- 'python tools/ab_bits.py --compare tools/_ab/libmxlo_r05.so linearoperators.jl_amd/csrc/libmxlo.so herm'
"""
import hashlib
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def digest(t):
    return hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()[:16]


def cases_herm(lo, torch, dev):
    for dt in (torch.float64, torch.float32):
        for n in (100, 256, 1000, 1024, 2048, 3000, 4096, 5000, 8192):
            g = torch.Generator(device="cpu").manual_seed(n)
            M = (torch.rand(n, n, dtype=dt, generator=g) - 0.5).to(dev).t()
            d, x, y = ((torch.rand(n, dtype=dt, generator=g) - 0.5).to(dev) for _ in range(3))
            H = lo.opHermitian(d, M)
            r = y.clone()
            lo.mul(r, H, x, 1.0, 0.0)
            yield f"herm {str(dt)[6:]} n={n} b0", r
            r = y.clone()
            lo.mul(r, H, x, 0.7, -1.3)
            yield f"herm {str(dt)[6:]} n={n} ab", r
            for k in (2, 3, 4, 7):
                V = (torch.rand(k, n, dtype=dt, generator=g) - 0.5).to(dev).t()
                R = torch.zeros(k, n, dtype=dt, device=dev).t()
                lo.mul(R, H, V, 1.0, 0.0)
                yield f"herm {str(dt)[6:]} n={n} block k={k}", R


def cases_dense(lo, torch, dev):
    for dt in (torch.float64, torch.float32):
        for (m, n) in ((1000, 700), (4096, 4096), (5000, 3000), (16384, 2048), (2048, 16384)):
            g = torch.Generator(device="cpu").manual_seed(m * 7 + n)
            M = (torch.rand(n, m, dtype=dt, generator=g) - 0.5).to(dev).t()
            op = lo.LinearOperatorFromMatrix(M)
            for k in (1, 2, 3, 4, 8, 11):
                V = (torch.rand(k, n, dtype=dt, generator=g) - 0.5).to(dev).t()
                R = (torch.rand(k, m, dtype=dt, generator=g) - 0.5).to(dev).t()
                if k == 1:
                    V, R = V[:, 0].contiguous(), R[:, 0].contiguous()
                r = R.clone()
                lo.mul(r, op, V, 1.0, 0.0)
                yield f"dense {str(dt)[6:]} {m}x{n} k={k} b0", r
                r = R.clone()
                lo.mul(r, op, V, 0.7, -1.3)
                yield f"dense {str(dt)[6:]} {m}x{n} k={k} ab", r
                U = (torch.rand(k, m, dtype=dt, generator=g) - 0.5).to(dev).t()
                Rt = torch.zeros(k, n, dtype=dt, device=dev).t()
                if k == 1:
                    U, Rt = U[:, 0].contiguous(), Rt[:, 0].contiguous()
                lo.mul(Rt, op.T, U, 1.0, 0.0)
                yield f"dense {str(dt)[6:]} {m}x{n} k={k} T", Rt


def cases_qn(lo, torch, dev):
    import numpy as np
    for dt in (torch.float64, torch.float32):
        for kind, ctor in (("inv", lo.InverseLBFGSOperator), ("fwd", lo.LBFGSOperator), ("sr1", lo.LSR1Operator)):
            for n, mem in ((1000, 5), (70001, 5), (1 << 20, 10), (3_000_017, 20)):
                rng = np.random.default_rng(n + mem)
                op = ctor(dt, n, mem=mem, device=dev)
                x = torch.from_numpy(rng.uniform(-1, 1, n)).to(dt).to(dev)
                for it in range(mem + 3):
                    s = torch.from_numpy(rng.uniform(-1, 1, n)).to(dt).to(dev)
                    yv = s * (1.0 + 0.5 * torch.from_numpy(rng.uniform(0, 1, n)).to(dt).to(dev))
                    if kind == "sr1":
                        yv = yv + 0.1 * torch.from_numpy(rng.uniform(-1, 1, n)).to(dt).to(dev)
                    lo.push(op, s, yv)
                r = torch.zeros(n, dtype=dt, device=dev)
                lo.mul(r, op, x, 1.0, 0.0)
                yield f"qn {kind} {str(dt)[6:]} n={n} m={mem} apply-after-pushes", r


def cases_persist(lo, torch, dev):
    """quasi-Newton applies in the range of the persistent single launch (qn.hip: qn_apply_persist_kernel), ragged lengths, alpha / beta / shift"""
    import numpy as np
    for dt in (torch.float64, torch.float32):
        for kind, ctor in (("inv", lo.InverseLBFGSOperator), ("fwd", lo.LBFGSOperator), ("sr1", lo.LSR1Operator)):
            for n, mem in (((1 << 19) + 5, 7), (1 << 19, 10), (700_001, 3), (1 << 20, 5), (1_500_001, 5), ((1 << 21) + 2, 10), (1 << 22, 5), (5_000_000, 4)):
                rng = np.random.default_rng(n + mem)
                op = ctor(dt, n, mem=mem, device=dev)
                x = torch.from_numpy(rng.uniform(-1, 1, n)).to(dt).to(dev)
                for it in range(mem + 2):
                    s = torch.from_numpy(rng.uniform(-1, 1, n)).to(dt).to(dev)
                    yv = s * (1.0 + 0.5 * torch.from_numpy(rng.uniform(0, 1, n)).to(dt).to(dev))
                    if kind == "sr1":
                        yv = yv + 0.1 * torch.from_numpy(rng.uniform(-1, 1, n)).to(dt).to(dev)
                    lo.push(op, s, yv)
                r = torch.from_numpy(rng.uniform(-1, 1, n)).to(dt).to(dev)
                lo.mul(r, op, x, 1.0, 0.0)
                yield f"persist {kind} {str(dt)[6:]} n={n} m={mem} b0", r
                lo.mul(r, op, x, 0.7, -1.3)
                yield f"persist {kind} {str(dt)[6:]} n={n} m={mem} ab", r
                del op


SECTIONS = {"herm": cases_herm, "dense": cases_dense, "qn": cases_qn, "persist": cases_persist}


def run(sections):
    import torch
    import __graft_entry__ as g
    lo = g.load_package()
    dev = torch.device("cuda", 0)
    for s in sections:
        for name, t in SECTIONS[s](lo, torch, dev):
            print(f"{digest(t)}  {name}", flush=True)


def compare(a, b, sections):
    outs = []
    for lib in (a, b):
        env = dict(os.environ, MXLO_LIB_PATH=os.path.abspath(lib))
        p = subprocess.run([sys.executable, os.path.abspath(__file__)] + sections, env=env, stdout=subprocess.PIPE, text=True)
        if p.returncode != 0:
            print(f"# run under {lib} failed (rc {p.returncode})")
            return 1
        outs.append({ln.split("  ", 1)[1]: ln.split("  ", 1)[0] for ln in p.stdout.splitlines() if "  " in ln})
    diff = [k for k in outs[0] if outs[0][k] != outs[1].get(k)]
    print(f"# {len(outs[0])} cases under A = {a} and B = {b}: {len(diff)} differ in bits")
    for k in diff:
        print(f"DIFF {k}: {outs[0][k]} vs {outs[1].get(k)}")
    return 1 if diff else 0


if __name__ == "__main__":
    args = sys.argv[1:]
    if args and args[0] == "--compare":
        sys.exit(compare(args[1], args[2], args[3:] or list(SECTIONS)))
    run(args or list(SECTIONS))
