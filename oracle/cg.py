"""Real-basis Clebsch–Gordan (Wigner-3j-normalised) tensors for l <= 2, computed numerically.
TEST INFRASTRUCTURE (+ dev-time generator of the product's committed tables, tools/gen_cg_tables.py).

Builder-defined convention (e3nn is unavailable offline; "e3nn 3j parity unpinned"):
  * l=0 basis: 1.   l=1 basis: (x, y, z) — the reference's vector basis (dot/cross at L1TP.py:247,279).
  * l=2 basis (orthonormal on the unit sphere up to the common factor, sum of squares = 1):
        b0 = sqrt3 xy,  b1 = sqrt3 yz,  b2 = (2z^2 - x^2 - y^2)/2,  b3 = sqrt3 zx,  b4 = (sqrt3/2)(x^2 - y^2)
  * C[l1,l2,l3][m1,m2,m3]: the unique invariant tensor of V_l1 (x) V_l2 (x) V_l3, unit Frobenius norm
    (the normalisation of the reference's constants: cg000 = 1, cg110 = cg011 = 1/sqrt3, cg111 = 1/sqrt6,
    L1TP.py:91-94).  Sign: chosen so that the l<=1 cases equal the reference exactly
    (C[1,1,0] = +delta/sqrt3, C[0,1,1] = C[1,0,1] = +delta/sqrt3, C[1,1,1] = +epsilon_ijk/sqrt6, i.e.
    out = in1 x in2); otherwise the first non-zero entry in C order (m1,m2,m3) is positive.
  * "component" spherical harmonics: Y0 = 1, Y1 = sqrt3 (x,y,z)/r, Y2 = sqrt5 b(r/|r|).
"""
import itertools

import numpy as np

S3 = np.sqrt(3.0)


def basis(l, r):
    """r [...,3] (need not be unit) -> homogeneous degree-l basis polynomials [..., 2l+1]."""
    x, y, z = r[..., 0], r[..., 1], r[..., 2]
    if l == 0:
        return np.ones(r.shape[:-1] + (1,))
    if l == 1:
        return np.stack([x, y, z], -1)
    if l == 2:
        return np.stack([S3 * x * y, S3 * y * z, (2 * z * z - x * x - y * y) / 2, S3 * z * x, S3 / 2 * (x * x - y * y)], -1)
    raise ValueError(l)


def sh_component(lmax, rel):
    """[..,3] -> [.., (lmax+1)^2] component-normalised real SH of the direction of `rel` (0 for rel = 0, l>0)."""
    rel = np.asarray(rel, dtype=np.float64)
    d = np.sqrt((rel * rel).sum(-1, keepdims=True))
    u = np.divide(rel, d, out=np.zeros_like(rel), where=d > 0)
    out = [np.ones(rel.shape[:-1] + (1,))]
    for l in range(1, lmax + 1):
        out.append(np.sqrt(2 * l + 1.0) * basis(l, u) * (d > 0))
    return np.concatenate(out, -1)


def rotation_matrices(l, R):
    """D_l(R) with basis(l, R r) = D_l(R) basis(l, r)."""
    if l == 0:
        return np.ones((1, 1))
    rng = np.random.default_rng(12345)
    pts = rng.normal(size=(64, 3))
    A = basis(l, pts)                 # [P, d]
    B = basis(l, pts @ R.T)           # [P, d]  = A @ D^T
    D = np.linalg.lstsq(A, B, rcond=None)[0].T
    return D


def random_rotation(rng):
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    return q * np.sign(np.linalg.det(q))


_cache = {}


def cg(l1, l2, l3):
    """-> C [2l1+1, 2l2+1, 2l3+1] or None when the triangle rule forbids the coupling."""
    key = (l1, l2, l3)
    if key in _cache:
        return _cache[key]
    if not (abs(l1 - l2) <= l3 <= l1 + l2):
        _cache[key] = None
        return None
    d1, d2, d3 = 2 * l1 + 1, 2 * l2 + 1, 2 * l3 + 1
    n = d1 * d2 * d3
    rng = np.random.default_rng(2024)
    M = np.zeros((n, n))
    for _ in range(6):
        R = random_rotation(rng)
        K = np.kron(np.kron(rotation_matrices(l1, R), rotation_matrices(l2, R)), rotation_matrices(l3, R))
        M += (K - np.eye(n)).T @ (K - np.eye(n))
    w, v = np.linalg.eigh(M)
    assert w[0] < 1e-10 and (n == 1 or w[1] > 1e-6), (key, w[:3])   # exactly one invariant
    C = v[:, 0].reshape(d1, d2, d3)
    C[np.abs(C) < 1e-12] = 0.0
    C /= np.sqrt((C * C).sum())
    # sign convention
    if key == (1, 1, 1):
        s = np.sign(C[0, 1, 2])
    else:
        s = np.sign(C.reshape(-1)[np.flatnonzero(C.reshape(-1))[0]])
    C = C * s
    _cache[key] = C
    return C


def all_tables(lmax=2):
    return {k: cg(*k) for k in itertools.product(range(lmax + 1), repeat=3) if cg(*k) is not None}
