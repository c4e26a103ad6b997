"""Robustness of the Gram / compact forward L-BFGS forms when the stored steps are nearly collinear (what happens
late in an optimisation): relative error of B*x against a dense BFGS recursion evaluated in np.longdouble, for the
three push modes and the reference-ordered oracle, as the spread `eps` of the steps around a common direction shrinks."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import __graft_entry__ as g
lo = g.load_package()
import oracle
dev = torch.device("cuda:0")
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
n, mem = 300, 8
for eps in (1.0, 1e-2, 1e-4, 1e-6, 1e-8):
    rng = np.random.default_rng(3)
    d0 = rng.standard_normal(n)
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    Hm = (Q * rng.uniform(0.5, 50.0, n)) @ Q.T                 # SPD Hessian, condition 100
    pairs = []
    for k in range(mem + 3):
        s = d0 * (1.0 if eps < 1 else 0.0) + eps * rng.standard_normal(n)
        pairs.append((s, Hm @ s))
    # truth: dense BFGS on the last `mem` pairs in extended precision, B0 = I/gamma (gamma from the newest pair)
    L = np.longdouble
    kept = pairs[-mem:]
    sN, yN = kept[-1]
    gamma = (L(1) * (yN.astype(L) @ sN.astype(L))) / (yN.astype(L) @ yN.astype(L))
    B = np.eye(n, dtype=L) / gamma
    for s, y in kept:
        s, y = s.astype(L), y.astype(L)
        Bs = B @ s
        B = B - np.outer(Bs, Bs) / (s @ Bs) + np.outer(y, y) / (y @ s)
    # x inside span(S): the part of B that the stored pairs actually shape (a random x is dominated by x/gamma)
    x = kept[-1][0] - kept[-2][0]                               # the eps-scale directions inside span(S)
    x /= np.linalg.norm(x)
    truth = (B @ x.astype(L)).astype(np.float64)
    corr = np.linalg.norm(truth - x / float(gamma))            # size of the part the stored pairs contribute
    errs = {"|corr|/|Bx|": corr / np.linalg.norm(truth)}
    O = oracle.LBFGS(n, mem=mem, inverse=False)
    for s, y in pairs:
        O.push(s, y)
    errs["oracle(ref order)"] = np.linalg.norm(O.mul(np.empty(n), x) - truth) / corr
    for mode in ("reforder", "gram", "compact"):
        op = lo.LBFGSOperator(n, mem=mem, device=dev).set_push_mode(mode)
        for s, y in pairs:
            lo.push(op, T(s), T(y))
        got = (op * T(x)).cpu().numpy()
        errs[mode] = np.linalg.norm(got - truth) / corr
    print(f"eps={eps:7.0e}  " + "  ".join(f"{k}: {v:9.2e}" for k, v in errs.items()), flush=True)

print("\ninverse operator (two-pass Gram form vs reference-ordered two-loop), error relative to |H x - gamma x|")
for eps in (1.0, 1e-2, 1e-4, 1e-6, 1e-8):
    rng = np.random.default_rng(3)
    d0 = rng.standard_normal(n)
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    Hm = (Q * rng.uniform(0.5, 50.0, n)) @ Q.T
    pairs = []
    for k in range(mem + 3):
        s = d0 * (1.0 if eps < 1 else 0.0) + eps * rng.standard_normal(n)
        pairs.append((s, Hm @ s))
    L = np.longdouble
    kept = pairs[-mem:]
    sN, yN = kept[-1]
    gamma = (yN.astype(L) @ sN.astype(L)) / (yN.astype(L) @ yN.astype(L))
    H = np.eye(n, dtype=L) * gamma
    I = np.eye(n, dtype=L)
    for s, y in kept:
        s, y = s.astype(L), y.astype(L)
        rho = 1 / (y @ s)
        V = I - rho * np.outer(s, y)
        H = V @ H @ V.T + rho * np.outer(s, s)
    x = kept[-1][1] - kept[-2][1]
    x /= np.linalg.norm(x)
    truth = (H @ x.astype(L)).astype(np.float64)
    corr = np.linalg.norm(truth - float(gamma) * x)
    errs = {"|corr|/|Hx|": corr / np.linalg.norm(truth)}
    O = oracle.LBFGS(n, mem=mem, inverse=True)
    for s, y in pairs:
        O.push(s, y)
    errs["oracle(ref order)"] = np.linalg.norm(O.mul(np.empty(n), x) - truth) / corr
    for mode in ("reforder", "twopass"):
        op = lo.InverseLBFGSOperator(n, mem=mem, device=dev).set_mode(mode)
        for s, y in pairs:
            lo.push(op, T(s), T(y))
        errs[mode] = np.linalg.norm((op * T(x)).cpu().numpy() - truth) / corr
    print(f"eps={eps:7.0e}  " + "  ".join(f"{k}: {v:9.2e}" for k, v in errs.items()), flush=True)
