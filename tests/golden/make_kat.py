"""Generate tests/golden/kat_reference_tests.json.

The reference (LinearOperators.jl v2.14.2) is pure Julia and Julia is not installed in the build
image, so its outputs cannot be recorded by running it. This script instead writes down the
KNOWN-ANSWER cases that the reference's OWN test-suite holds for the `mul!` hot path: the
deterministic inputs each test constructs (`simple_vector(T,n) = [1,-1,1,...]`, test/test_aux.jl:33;
`s = i*ones(n)`, `y = [i; ones(n-1)]`, test/test_lbfgs.jl:35-36) and the closed-form value the test
asserts (`D*u == v.*u`, `P*v == v[idx]`, `Matrix(LB) ≈ dense BFGS`, ...). Expected values are
evaluated here in EXACT rational arithmetic (fractions.Fraction) from those closed forms — e.g. the
L-BFGS answers come from the dense BFGS / inverse-BFGS / SR1 rank updates that
test/test_lbfgs.jl:73-99 and test/test_lsr1.jl:43-68 compare against, not from any restatement of
the limited-memory recursions. Each case cites the reference test lines it encodes.

Run:  python tests/golden/make_kat.py   (rewrites the JSON next to this file)
"""
import json
import os
from fractions import Fraction as F

HERE = os.path.dirname(os.path.abspath(__file__))


def simple_vector(n):  # test/test_aux.jl:33
    return [F(-((-1) ** i)) for i in range(1, n + 1)]


def fl(v):
    return [float(x) for x in v]


def matvec(M, v):
    return [sum(M[i][j] * v[j] for j in range(len(v))) for i in range(len(M))]


def outer(a, b):
    return [[x * y for y in b] for x in a]


def dotf(a, b):
    return sum(x * y for x, y in zip(a, b))


def eye(n):
    return [[F(int(i == j)) for j in range(n)] for i in range(n)]


cases = []

# ---------------------------------------------------------------- opDiagonal  (test_linop.jl:308-319)
n = 10
v = simple_vector(n)
u = simple_vector(n)
res0 = simple_vector(n)
cases.append(dict(
    name="opDiagonal_square", ref="test/test_linop.jl:308-319", kind="diag",
    d=fl(v), u=fl(u), expect_apply=fl([a * b for a, b in zip(v, u)]),          # D*u == v.*u
    alpha=2.0, beta=2.0, res0=fl(res0),
    expect_mul5=fl([a * b * 2 + 2 * r for a, b, r in zip(v, u, res0)]),          # v.*u.*2 + 2 .* res
    tol="bit-exact (small integers)"))

# ---------------------------------------------------------------- rectangular opDiagonal (test_linop.jl:321-344)
nmax, nmin = 10, 6
vd = simple_vector(nmin)
uu = simple_vector(nmin)
ww = simple_vector(nmax)
cases.append(dict(
    name="opDiagonal_tall", ref="test/test_linop.jl:321-335", kind="diag_rect", nrow=nmax, ncol=nmin,
    d=fl(vd), u=fl(uu), expect_apply=fl([a * b for a, b in zip(vd, uu)] + [F(0)] * (nmax - nmin)),
    w=fl(ww), expect_tapply=fl([a * b for a, b in zip(vd, ww[:nmin])])))
cases.append(dict(
    name="opDiagonal_wide", ref="test/test_linop.jl:337-344", kind="diag_rect", nrow=nmin, ncol=nmax,
    d=fl(vd), u=fl(ww), expect_apply=fl([a * b for a, b in zip(vd, ww[:nmin])]),
    w=fl(uu), expect_tapply=fl([a * b for a, b in zip(vd, uu)] + [F(0)] * (nmax - nmin))))

# ---------------------------------------------------------------- opHouseholder (test_linop.jl:511-518)
hv = simple_vector(n)
hu = simple_vector(n)
c = 2 * dotf(hv, hu)
cases.append(dict(
    name="opHouseholder", ref="test/test_linop.jl:511-518", kind="householder",
    h=fl(hv), u=fl(hu), expect_apply=fl([x - c * h for x, h in zip(hu, hv)])))   # u - 2*dot(v,u)*v = -19u

# ---------------------------------------------------------------- restriction / extension (test_linop.jl:437-461)
rv = simple_vector(10)
for nm, idx in (("J", [1, 2, 4, 7]), ("r", list(range(3, 7))), ("s", list(range(1, 8, 2))), ("k", [4])):
    w = [rv[i - 1] for i in idx]
    vz = [F(0)] * 10
    for i in idx:
        vz[i - 1] = rv[i - 1]
    spec = {"J": {"list": idx}, "r": {"range": [3, 6, 1]}, "s": {"range": [1, 7, 2]}, "k": {"scalar": 4}}[nm]
    cases.append(dict(name=f"restriction_{nm}", ref="test/test_linop.jl:437-461", kind="restriction", n=10,
                      idx=spec, v=fl(rv), expect_w=fl(w), expect_vz=fl(vz)))
cases.append(dict(name="restriction_colon", ref="test/test_linop.jl:445 (idx = Colon())", kind="restriction", n=10,
                  idx={"colon": True}, v=fl(rv), expect_w=fl(rv), expect_vz=fl(rv)))

# ---------------------------------------------------------------- cat with opEye/opZeros (test_cat.jl:43-49)
cases.append(dict(name="cat_eye_zeros", ref="test/test_cat.jl:44-46", kind="cat_eye_zeros",
                  v=fl(simple_vector(5)), expect=fl(simple_vector(5))))
cases.append(dict(name="cat_vcat_eye", ref="test/test_cat.jl:48-50", kind="cat_vcat_eye",
                  v=fl(simple_vector(2)), expect=fl(simple_vector(2) + simple_vector(2))))

# ---------------------------------------------------------------- solve_shifted_system! / ldiv! (test_solve_shifted_system.jl:5-61)
# setup_test_val: B = LBFGSOperator(n, mem = M, scaling), 10 pushes, x known, b = B*x + sigma*x; the test asserts
# solve_shifted_system!(x_sol, B, b, sigma) ≈ x (1e-6) and, for sigma = 0, ldiv!(x_sol, B, b) ≈ H*b with the inverse
# operator. The test draws rand pairs; here the pairs are deterministic rationals with positive curvature and the dense
# BFGS matrix of the last M pairs (B0 = I / gamma, gamma = y's/y'y of the LAST pair when scaling) is formed exactly.
def solve_exact(Mx, rhs):
    nn = len(rhs)
    Aug = [list(Mx[i]) + [rhs[i]] for i in range(nn)]
    for c_ in range(nn):
        piv = next(r for r in range(c_, nn) if Aug[r][c_] != 0)
        Aug[c_], Aug[piv] = Aug[piv], Aug[c_]
        for r in range(nn):
            if r != c_ and Aug[r][c_] != 0:
                f_ = Aug[r][c_] / Aug[c_][c_]
                Aug[r] = [a - f_ * b_ for a, b_ in zip(Aug[r], Aug[c_])]
    return [Aug[i][nn] / Aug[i][i] for i in range(nn)]


ns, Ms = 8, 5
spairs = []
for k in range(1, 11):
    sv = [F(((i + 1) * k) % 7 + 1, 4) * (1 if (i + k) % 3 else -1) for i in range(ns)]
    yv = [sv[i] * (1 + F(i % 3, 2)) + F((i + k) % 5 - 2, 16) for i in range(ns)]
    assert dotf(sv, yv) > 0
    spairs.append((sv, yv))
xs_true = [F((-1) ** i * (i + 1), 2) for i in range(ns)]
for scaling_ in (False, True):
    kept_ = spairs[-Ms:]
    gam = dotf(kept_[-1][1], kept_[-1][0]) / dotf(kept_[-1][1], kept_[-1][1]) if scaling_ else F(1)
    Bd = [[(1 / gam if i == j else F(0)) for j in range(ns)] for i in range(ns)]
    Hd = eye(ns)
    for sv, yv in kept_:
        Bs_ = matvec(Bd, sv)
        Bd = [[Bd[i][j] - Bs_[i] * Bs_[j] / dotf(sv, Bs_) + yv[i] * yv[j] / dotf(yv, sv) for j in range(ns)] for i in range(ns)]
        rho_ = 1 / dotf(yv, sv)
        Hy_ = matvec(Hd, yv)
        Hd = [[Hd[i][j] - rho_ * (sv[i] * Hy_[j] + Hy_[i] * sv[j]) + rho_ * (1 + rho_ * dotf(yv, Hy_)) * sv[i] * sv[j]
               for j in range(ns)] for i in range(ns)]
    for sig in ((F(1, 8), F(0), F(3)) if not scaling_ else (F(1, 8), F(2))):
        bvec = [a + sig * x_ for a, x_ in zip(matvec(Bd, xs_true), xs_true)]
        # the exact solution of (B + sigma I) x = float(b): b is rounded to Float64 when stored, so solve for THAT b
        bfl = [F(float(t)) for t in bvec]
        Bsig = [[Bd[i][j] + (sig if i == j else 0) for j in range(ns)] for i in range(ns)]
        case = dict(name=f"solve_shifted_scaling{int(scaling_)}_sigma{float(sig):g}",
                    ref="test/test_solve_shifted_system.jl:5-61 (deterministic pairs)", kind="solve_shifted",
                    n=ns, mem=Ms, scaling=scaling_, sigma=float(sig), pairs=[dict(s=fl(a), y=fl(b_)) for a, b_ in spairs],
                    b=[float(t) for t in bvec], expect_x=fl(solve_exact(Bsig, bfl)), x_true=fl(xs_true),
                    tol="1e-10 relative (reference bar: isapprox(atol = rtol = 1e-6))")
        if sig == 0 and not scaling_:
            case["expect_Hb"] = fl(matvec(Hd, bfl))            # ldiv! test: x == H*b with the inverse operator (:49-60)
        cases.append(case)

# ---------------------------------------------------------------- diagonal quasi-Newton apply (test_diag.jl:75-106)
for nm, d in (("DiagonalPSB_gradf", [2, -1, 2]), ("DiagonalAndrei_gradf", [2, -2, 2])):
    x = [F(3), F(-5), F(7)]
    cases.append(dict(name=nm, ref="test/test_diag.jl:75-101 (B.d) + src/DiagonalHessianApproximation.jl:37,112",
                      kind="diag", d=[float(t) for t in d], u=fl(x), expect_apply=fl([F(a) * b for a, b in zip(d, x)]),
                      alpha=2.0, beta=2.0, res0=fl(x), expect_mul5=fl([F(a) * b * 2 + 2 * b for a, b in zip(d, x)])))
cases.append(dict(name="SpectralGradient_gradf", ref="test/test_diag.jl:91,103-105 (sigma = 2, 1-element d)",
                  kind="diag_scalar", d=[2.0], u=fl([F(3), F(-5), F(7)]), expect_apply=[6.0, -10.0, 14.0]))

# ---------------------------------------------------------------- L-BFGS (test_lbfgs.jl:7-56, 73-99)
n, mem = 10, 5
pairs = []
for i in range(1, mem + 3):                       # test_lbfgs.jl:33-43
    s = [F(i)] * n
    y = [F(i)] + [F(1)] * (n - 1)
    if dotf(s, y) > 0:
        pairs.append((s, y))
kept = pairs[-mem:]                               # circular buffer keeps the last `mem` pairs
B = eye(n)
H = eye(n)
for s, y in kept:                                 # dense BFGS (test_lbfgs.jl:77-86) and its inverse
    Bs = matvec(B, s)
    sBs = dotf(s, Bs)
    ys = dotf(y, s)
    B = [[B[i][j] - Bs[i] * Bs[j] / sBs + y[i] * y[j] / ys for j in range(n)] for i in range(n)]
    rho = 1 / ys
    Hy = matvec(H, y)
    yHy = dotf(y, Hy)
    # H+ = H - rho (s Hy' + Hy s') + rho (1 + rho yHy) s s'
    H = [[H[i][j] - rho * (s[i] * Hy[j] + Hy[i] * s[j]) + rho * (1 + rho * yHy) * s[i] * s[j]
          for j in range(n)] for i in range(n)]
vv = simple_vector(n)
cases.append(dict(
    name="LBFGS_mem5_7pushes", ref="test/test_lbfgs.jl:7-56 (pairs :33-43) vs dense BFGS :73-99", kind="lbfgs",
    n=n, mem=mem, scaling=False,
    pre_rejected=[dict(s=fl(vv), y=fl([-x for x in vv])), dict(s=fl(vv), y=[0.0] * n)],   # :24-31 insert stays 1
    pairs=[dict(s=fl(s), y=fl(y)) for s, y in pairs],
    expect_insert=len(pairs) % mem + 1,                                                  # :45-46
    expect_ys_slots=fl([dotf(y, s) for s, y in (pairs[5], pairs[6], pairs[2], pairs[3], pairs[4])]),
    v=fl(vv), expect_Bv=fl(matvec(B, vv)), expect_Hv=fl(matvec(H, vv)),
    expect_diagB=fl([B[i][i] for i in range(n)]),                                          # :14,54
    tol="1e-12 relative (reference bar: sqrt(eps))"))

# full-memory BFGS with identical pairs (test_lbfgs.jl:88-96): s = y = simple_vector -> B stays I
cases.append(dict(name="LBFGS_fullmem_identity_pairs", ref="test/test_lbfgs.jl:73-99", kind="lbfgs_identity",
                  n=n, mem=n, scaling=False, pairs=[dict(s=fl(vv), y=fl(vv)) for _ in range(n)],
                  v=fl([F(i) for i in range(1, n + 1)]), expect_Bv=fl([F(i) for i in range(1, n + 1)])))

# ---------------------------------------------------------------- L-SR1 (test_lsr1.jl:6-28, 43-68)
B = eye(n)
acc = []
for i in range(1, mem + 3):
    s = [F(i)] * n
    y = [F(i)] + [F(1)] * (n - 1)
    acc.append((s, y))
# replay with the limited-memory semantics: an update is rejected when (y-Bs)'s == 0 (well_defined,
# src/lsr1.jl:131); accepted pairs are kept in a circular buffer of `mem`, B rebuilt from B0 = I.
stored = []
for s, y in acc:
    Bk = eye(n)
    for (ss, yy) in stored[-mem:]:
        r = [a - b for a, b in zip(yy, matvec(Bk, ss))]
        dn = dotf(r, ss)
        Bk = [[Bk[i][j] + r[i] * r[j] / dn for j in range(n)] for i in range(n)]
    r = [a - b for a, b in zip(y, matvec(Bk, s))]
    if dotf(r, s) != 0:
        stored.append((s, y))
Bk = eye(n)
for (ss, yy) in stored[-mem:]:
    r = [a - b for a, b in zip(yy, matvec(Bk, ss))]
    dn = dotf(r, ss)
    Bk = [[Bk[i][j] + r[i] * r[j] / dn for j in range(n)] for i in range(n)]
cases.append(dict(
    name="LSR1_mem5_7pushes", ref="test/test_lsr1.jl:6-28 vs dense SR1 :43-68", kind="lsr1", n=n, mem=mem, scaling=False,
    pairs=[dict(s=fl(s), y=fl(y)) for s, y in acc], expect_naccepted=len(stored),
    expect_insert=len(stored) % mem + 1, v=fl(vv), expect_Bv=fl(matvec(Bk, vv)),
    expect_diagB=fl([Bk[i][i] for i in range(n)]), tol="1e-12 relative"))

# ---------------------------------------------------------------- diagonal quasi-Newton push! (test_diag.jl:75-106)
# x0 = [-1,1,-1], x1 = x0 + [1,0,1]; s = x1 - x0; y = grad(x1) - grad(x0); B = DQN([1,-1,1]); push!(B, s, y);
# the test holds B.d (Bref, :79-93) for DiagonalPSB / DiagonalAndrei and B.d[1] for SpectralGradient(1.0, 3).
import math
_s = [1.0, 0.0, 1.0]
_g1 = math.sin(-1.0) - math.exp(-1.0)
for gname, y, psb, andrei, spg, tol in (
        ("gradf", [2.0, 0.0, 2.0], [2.0, -1.0, 2.0], [2.0, -2.0, 2.0], 2.0, 0.0),
        ("gradg", [1.0 - math.exp(-1.0), 0.0, math.sin(-1.0)], [1 + (_g1 - 1) / 2, -1.0, 1 + (_g1 - 1) / 2],
         [(1 + _g1) / 2, -2.0, (1 + _g1) / 2], (1 - math.exp(-1.0) + math.sin(-1.0)) / 2, 1e-10),
        ("gradh", [-2.0, 1.0, -3.0], [-2.5, -1.0, -2.5], [-2.5, -2.0, -2.5], -2.5, 0.0)):
    cases.append(dict(name=f"diagqn_push_{gname}", ref="test/test_diag.jl:41-50 (points, gradients), :75-106 (Bref)",
                      kind="diagqn_push", d0=[1.0, -1.0, 1.0], s=_s, y=y, expect_psb=psb, expect_andrei=andrei,
                      expect_spectral=spg, tol=max(tol, 1e-15)))

# ================================================================ round 2: complex, Hermitian, kron
# Gaussian rationals as (re, im) Fractions; JSON carries [re, im] pairs.
def cmul(a, b):
    return (a[0] * b[0] - a[1] * b[1], a[0] * b[1] + a[1] * b[0])


def cadd(a, b):
    return (a[0] + b[0], a[1] + b[1])


def csub(a, b):
    return (a[0] - b[0], a[1] - b[1])


def cconj(a):
    return (a[0], -a[1])


def cfl(v):
    return [[float(x[0]), float(x[1])] for x in v]


def cdot(a, b):            # LinearAlgebra.dot conjugates its FIRST argument
    acc = (F(0), F(0))
    for x, y in zip(a, b):
        acc = cadd(acc, cmul(cconj(x), y))
    return acc


def csimple(n):            # simple_vector(ComplexF64, n): the same +1/-1 pattern, zero imaginary part
    return [(F(-((-1) ** i)), F(0)) for i in range(1, n + 1)]


def cpattern(n, a, b):     # a deterministic vector with non-trivial imaginary parts (dyadic: exact in binary)
    return [(F(-((-1) ** i)) * F(a + i, 4), F(((-1) ** (i // 2)) * (b + 2 * i), 8)) for i in range(1, n + 1)]


n = 10
two = (F(2), F(0))
for nm, cv, cu, cr in (("simple", csimple(n), csimple(n), csimple(n)),
                       ("dyadic", cpattern(n, 1, 3), cpattern(n, 2, 5), cpattern(n, 3, 1))):
    # test/test_linop.jl:308-318 on ComplexF64 inputs (round 1 had silently re-typed this case to Float64):
    #   D*u == v.*u, transpose(D)*u == v.*u, D'*u == conj(v).*u, mul!(res, D, u, 2.0, 2.0) == v.*u.*2.0 + 2.0.*res
    vu = [cmul(a, b) for a, b in zip(cv, cu)]
    cases.append(dict(
        name=f"opDiagonal_complex_{nm}", ref="test/test_linop.jl:308-318 (ComplexF64)", kind="cdiag",
        d=cfl(cv), u=cfl(cu), expect_apply=cfl(vu), expect_tapply=cfl(vu),
        expect_ctapply=cfl([cmul(cconj(a), b) for a, b in zip(cv, cu)]),
        alpha=2.0, beta=2.0, res0=cfl(cr),
        expect_mul5=cfl([cadd(cmul(two, x), cmul(two, r)) for x, r in zip(vu, cr)]),
        tol="bit-exact (dyadic rationals)"))
    # test/test_linop.jl:511-517: H*u == u - 2*dot(v,u)*v ; transpose(H)*u == u - 2*dot(conj(v),u)*conj(v) ; H'*u == H*u
    c2 = cmul(two, cdot(cv, cu))
    Hu = [csub(x, cmul(c2, h)) for x, h in zip(cu, cv)]
    vbar = [cconj(h) for h in cv]
    c2t = cmul(two, cdot(vbar, cu))
    Htu = [csub(x, cmul(c2t, h)) for x, h in zip(cu, vbar)]
    cases.append(dict(
        name=f"opHouseholder_complex_{nm}", ref="test/test_linop.jl:511-517 (ComplexF64)", kind="chouseholder",
        h=cfl(cv), u=cfl(cu), expect_apply=cfl(Hu), expect_tapply=cfl(Htu), expect_ctapply=cfl(Hu),
        tol="1e-12 relative (dot order unpinned)"))

# ---------------------------------------------------------------- opHermitian (test_linop.jl:360-380, real instantiation)
# C = tril(A,-1) + tril(A,-1)' + diagm(d); H = opHermitian(d, A): H*v == C*v, transpose(H)*v == transpose(C)*v,
# H'*v == C*v; opHermitian(C) for a symmetric C. The test draws A with rand; here A[i][j] = (3i - 2j + 1)/8 (dyadic).
nh = 7
Ah = [[F(3 * (i + 1) - 2 * (j + 1) + 1, 8) for j in range(nh)] for i in range(nh)]
dh = [F(i + 1, 2) - F(3, 4) for i in range(nh)]
Ch = [[(Ah[i][j] if i > j else (Ah[j][i] if j > i else dh[i])) for j in range(nh)] for i in range(nh)]
vh = simple_vector(nh)
xh = [F((-1) ** i * (i + 2), 4) for i in range(nh)]
r0h = [F(i - 3, 2) for i in range(nh)]
cases.append(dict(
    name="opHermitian_d_A", ref="test/test_linop.jl:360-370 (Float64 instantiation, deterministic A)", kind="hermitian",
    n=nh, A=[fl(r) for r in Ah], d=fl(dh), v=fl(vh), expect_apply=fl(matvec(Ch, vh)),
    x=fl(xh), alpha=1.5, beta=-0.5, res0=fl(r0h),
    expect_mul5=fl([F(3, 2) * a + F(-1, 2) * r for a, r in zip(matvec(Ch, xh), r0h)]),
    tol="1e-13 relative (exact in binary up to summation order)"))

# ---------------------------------------------------------------- opHermitian on ComplexF64 (test_linop.jl:360-370)
# A = simple_matrix(ComplexF64, n, n); d = real.(diag(A)); A = tril(A, -1); C = A + A' + diagm(0 => d); H = opHermitian(d, A)
# H*v == C*v, transpose(H)*v == transpose(C)*v, H'*v == C*v with v = simple_vector(ComplexF64, n) — and a second v with
# imaginary parts. The test draws A with rand; here A[i][j] = ((3i - 2j + 1)/8, (i + 2j - 3)/8) (Gaussian dyadic).
def cmatvec(Mx, x):
    out = []
    for row in Mx:
        acc = (F(0), F(0))
        for a, b in zip(row, x):
            acc = cadd(acc, cmul(a, b))
        out.append(acc)
    return out


nc = 7
Ac = [[(F(3 * (i + 1) - 2 * (j + 1) + 1, 8), F((i + 1) + 2 * (j + 1) - 3, 8)) for j in range(nc)] for i in range(nc)]
dc = [F(i + 1, 2) - F(3, 4) for i in range(nc)]                         # real.(diag(.)): a REAL diagonal
zero = (F(0), F(0))
Cc = [[(Ac[i][j] if i > j else (cconj(Ac[j][i]) if j > i else (dc[i], F(0)))) for j in range(nc)] for i in range(nc)]
Cct = [[Cc[j][i] for j in range(nc)] for i in range(nc)]                # transpose(C) (no conjugation)
for nm, vc in (("simple", csimple(nc)), ("dyadic", cpattern(nc, 2, 5))):
    r0c = cpattern(nc, 3, 1)
    al, be = (F(3, 2), F(-1)), (F(-1, 2), F(1, 4))                      # complex alpha, beta
    cases.append(dict(
        name=f"opHermitian_complex_{nm}", ref="test/test_linop.jl:360-370 (ComplexF64, deterministic A)", kind="chermitian",
        n=nc, A=[cfl(r) for r in Ac], d=fl(dc), v=cfl(vc), expect_apply=cfl(cmatvec(Cc, vc)),
        expect_tapply=cfl(cmatvec(Cct, vc)), expect_ctapply=cfl(cmatvec(Cc, vc)),
        alpha=[float(al[0]), float(al[1])], beta=[float(be[0]), float(be[1])], res0=cfl(r0c),
        expect_mul5=cfl([cadd(cmul(al, a), cmul(be, r)) for a, r in zip(cmatvec(Cc, vc), r0c)]),
        C=[cfl(r) for r in Cc], tol="1e-13 relative (exact in binary up to summation order)"))

# ---------------------------------------------------------------- kron (test_kron.jl:9-36, Float64 factors)
# K = kron(A, B) (Base.kron); T*x == K*x, T'*x == K'*x, transpose(T)*x == transpose(K)*x for 2x3 factors.
Ak = [[F(1), F(-1, 2), F(3, 4)], [F(2), F(1, 4), F(-5, 8)]]
Bk = [[F(1, 2), F(3), F(-1)], [F(-7, 4), F(1, 8), F(2)]]
mk, nk, pk, qk = 2, 3, 2, 3
K = [[Ak[i // pk][j // qk] * Bk[i % pk][j % qk] for j in range(nk * qk)] for i in range(mk * pk)]
xk = simple_vector(nk * qk)
xtk = simple_vector(mk * pk)
Kt = [[K[i][j] for i in range(mk * pk)] for j in range(nk * qk)]
r0k = [F(i, 2) for i in range(mk * pk)]
cases.append(dict(
    name="kron_2x3_2x3", ref="test/test_kron.jl:9-36 (Float64 factors, deterministic)", kind="kron",
    A=[fl(r) for r in Ak], B=[fl(r) for r in Bk], x=fl(xk), expect_apply=fl(matvec(K, xk)),
    xt=fl(xtk), expect_tapply=fl(matvec(Kt, xtk)), alpha=2.0, beta=3.0, res0=fl(r0k),
    expect_mul5=fl([2 * a + 3 * r for a, r in zip(matvec(K, xk), r0k)]), K=[fl(r) for r in K],
    tol="1e-12 * norm(K, 1) as test_kron.jl:35"))

# ---------------------------------------------------------------- kron(Float64 A, ComplexF64 B) (test_kron.jl:3-8)
# the reference loops B over simple_matrix(Float64, 2, 3), simple_matrix(ComplexF64, 2, 3), ...: the complex-B pairing,
# with deterministic Gaussian-dyadic entries; T*x, transpose(T)*x, T'*x and Matrix(T) against Base.kron.
Bkc = [[(F(1, 2), F(-3, 4)), (F(3), F(1, 8)), (F(-1), F(2))], [(F(-7, 4), F(1, 2)), (F(1, 8), F(-1)), (F(2), F(5, 8))]]
Kc = [[cmul((Ak[i // pk][j // qk], F(0)), Bkc[i % pk][j % qk]) for j in range(nk * qk)] for i in range(mk * pk)]
Kct = [[Kc[i][j] for i in range(mk * pk)] for j in range(nk * qk)]
Kch = [[cconj(Kc[i][j]) for i in range(mk * pk)] for j in range(nk * qk)]
xkc, xtkc = cpattern(nk * qk, 1, 3), cpattern(mk * pk, 2, 5)
r0kc = cpattern(mk * pk, 3, 1)
alk, bek = (F(2), F(-1, 2)), (F(3), F(1))
cases.append(dict(
    name="kron_real_2x3_complex_2x3", ref="test/test_kron.jl:3-36 (Float64 A, ComplexF64 B, deterministic)", kind="ckron",
    A=[fl(r) for r in Ak], B=[cfl(r) for r in Bkc], x=cfl(xkc), expect_apply=cfl(cmatvec(Kc, xkc)),
    xt=cfl(xtkc), expect_tapply=cfl(cmatvec(Kct, xtkc)), expect_ctapply=cfl(cmatvec(Kch, xtkc)),
    alpha=[float(alk[0]), float(alk[1])], beta=[float(bek[0]), float(bek[1])], res0=cfl(r0kc),
    expect_mul5=cfl([cadd(cmul(alk, a), cmul(bek, r)) for a, r in zip(cmatvec(Kc, xkc), r0kc)]), K=[cfl(r) for r in Kc],
    tol="1e-12 * norm(K, 1) as test_kron.jl:35"))

# ---------------------------------------------------------------- dense LinearOperator(A), LITERAL numbers (test_linop.jl:587-595)
# "Issue #80 / Test mul!": A = [1.0 1.0; 1.0 0.0]; op = LinearOperator(A); mul!(y, op, ones(2)); @test y == [2.0; 1.0]
# (a symmetric A: transpose(op) and op' give the same numbers; the 5-arg form follows src/constructors.jl:19-29).
cases.append(dict(
    name="dense_issue80_literal", ref="test/test_linop.jl:587-595 (literal A, x and expected y)", kind="dense",
    A=[[1.0, 1.0], [1.0, 0.0]], x=[1.0, 1.0], expect_apply=[2.0, 1.0], alpha=2.0, beta=-1.0, res0=[0.5, -3.0],
    expect_mul5=[2.0 * 2.0 - 0.5, 2.0 * 1.0 + 3.0], tol="exact"))

out = dict(
    about="Known-answer cases held by LinearOperators.jl v2.14.2's own tests for the mul! hot path; "
          "generated by tests/golden/make_kat.py (exact rational arithmetic, no reference code executed).",
    cases=cases)
with open(os.path.join(HERE, "kat_reference_tests.json"), "w") as f:
    json.dump(out, f, indent=1)
print(f"wrote {len(cases)} cases")
