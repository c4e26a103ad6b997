"""-m gpu: vectors longer than 2^31 elements (SURVEY §8a "maximum sizes": every index in the kernels and in the
ABI is 64-bit). Data is generated on the device; parity is checked (a) bit-exactly against the oracle on windows
placed at the start, across the 2^31 element boundary and at the very end (elementwise / indexing leaves are
local), and (b) through size-independent properties for the leaves with a global reduction (Householder is an
involution for ||h|| = 1; the inverse L-BFGS operator satisfies the secant equation H*y = s for the newest pair).
fp32 keeps the footprint at 8.6 GB per vector."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu

N = (1 << 31) + 12_345
W = 4096


def windows(n):
    return [(0, W), ((1 << 31) - W // 2, (1 << 31) + W // 2), (n - W, n), (n - 3, n)]


@pytest.fixture(scope="module")
def big(dev):
    free, _ = torch.cuda.mem_get_info(dev)
    if free < 90 * (1 << 30):
        pytest.skip("needs ~90 GB of free HBM")
    g = torch.Generator(device=dev)
    g.manual_seed(1234)
    mk = lambda: torch.rand(N, dtype=torch.float32, device=dev, generator=g) * 2 - 1
    return {"d": mk(), "v": mk(), "r": mk()}


def host(t, a, b):
    return t[a:b].cpu().numpy().copy()


def test_diag_eye_scale_windows_bit_exact(lo, dev, big):
    d, v, r0 = big["d"], big["v"], big["r"]
    D = lo.opDiagonal(d)
    res = r0.clone()
    lo.mul(res, D, v, 2.0, -3.0)                      # Float64 scalars on Float32 data -> MXLO_SCALARS_F64
    for a, b in windows(N):
        ref = oracle.diag_mul(host(r0, a, b), host(d, a, b), host(v, a, b), 2.0, -3.0, flags=oracle.SCALARS_F64)
        assert np.array_equal(host(res, a, b), ref), (a, b)
    res = torch.full_like(v, float("nan"))
    lo.mul(res, D, v)                                 # 3-arg form: pure fp32, beta == 0 never reads res
    for a, b in windows(N):
        ref = oracle.diag_mul(np.full(b - a, np.nan, np.float32), host(d, a, b), host(v, a, b), 1.0, 0.0)
        assert np.array_equal(host(res, a, b), ref), (a, b)
    E = lo.opEye(torch.float32, N, S=lo.storage_of(v))
    res = r0.clone()
    lo.mul(res, E, v, 0.5, 2.0)
    for a, b in windows(N):
        ref = oracle.eye_mul(host(r0, a, b), host(v, a, b), 0.5, 2.0, flags=oracle.SCALARS_F64 | oracle.TAIL_BETA)
        assert np.array_equal(host(res, a, b), ref), (a, b)


def test_restriction_extension_beyond_2_31(lo, dev, big):
    v = big["v"]
    S = lo.storage_of(v)
    lo_, hi_ = (1 << 31) - 1000, N - 7                # 1-based UnitRange straddling 2^31
    R = lo.opRestriction(lo.jrange(lo_, hi_), N, S=S)
    out = R * v
    assert out.numel() == hi_ - lo_ + 1
    assert torch.equal(out, v[lo_ - 1:hi_])
    St = lo.opRestriction(lo.jrange(N - 5000, N, 7), N, S=S)          # StepRange near the end
    assert torch.equal(St * v, v[N - 5001:N:7])
    idx = torch.tensor([N, 1, (1 << 31) + 1, (1 << 31), N - 1, 2], dtype=torch.int64)
    Rv = lo.opRestriction(idx, N, S=S)
    assert torch.equal(Rv * v, v[(idx - 1).to(dev)])
    u = torch.arange(1, idx.numel() + 1, dtype=torch.float32, device=dev)
    ext = Rv.T * u                                    # res .= 0; res[I] = u  over 2^31+ elements
    assert ext.numel() == N
    assert torch.equal(ext[(idx - 1).to(dev)], u)
    assert float(ext.sum().item()) == float(u.sum().item())          # everything else is exactly zero
    del ext, out


def test_householder_involution(lo, dev, big):
    h, v = big["d"], big["v"]
    # ||h||_2 = 1 with the norm accumulated in float64 chunks (no 17 GB temporaries)
    nrm2 = 0.0
    step = 1 << 28
    for a in range(0, N, step):
        nrm2 += float((h[a:a + step].double() ** 2).sum().item())
    h = h * float(1.0 / np.sqrt(nrm2))
    H = lo.opHouseholder(h)
    w = H * v
    back = H * w
    num = den = 0.0
    for a in range(0, N, step):
        num += float(((back[a:a + step].double() - v[a:a + step].double()) ** 2).sum().item())
        den += float((v[a:a + step].double() ** 2).sum().item())
    assert np.sqrt(num / den) <= 1e-5
    # elementwise check of one apply on the windows with c = 2*dot(h, v) accumulated in float64
    dot = 0.0
    for a in range(0, N, step):
        dot += float((h[a:a + step].double() * v[a:a + step].double()).sum().item())
    for a, b in windows(N):
        ref = host(v, a, b).astype(np.float64) - 2.0 * dot * host(h, a, b).astype(np.float64)
        got = host(w, a, b).astype(np.float64)
        assert np.max(np.abs(got - ref)) <= 1e-5 * max(1.0, np.max(np.abs(ref)))
    del w, back, h


def test_inverse_lbfgs_secant_at_2_31(lo, dev, big):
    d, s = big["d"], big["v"]
    op = lo.InverseLBFGSOperator(torch.float32, N, mem=2, scaling=True, device=dev)
    y = s * (d.abs() + 0.5)                            # y = D∘s with D in [0.5, 1.5]: ys > 0
    lo.push(op, s, y)
    s2 = big["r"]
    y2 = s2 * (d.abs() + 0.5)
    lo.push(op, s2, y2)
    Hy = op * y2                                       # BFGS secant equation for the newest pair
    step = 1 << 28
    num = den = 0.0
    for a in range(0, N, step):
        num += float(((Hy[a:a + step].double() - s2[a:a + step].double()) ** 2).sum().item())
        den += float((s2[a:a + step].double() ** 2).sum().item())
    assert np.sqrt(num / den) <= 1e-4
