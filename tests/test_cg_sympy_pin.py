"""Independent pin of the coupling tensors: `oracle/cg.py` builds them numerically as the invariant of
D_l1 (x) D_l2 (x) D_l3, and the product's `csrc/cg_tables.h` is generated from it -- one source.  Here the same tensors
come from sympy's exact Wigner 3j symbols (Condon-Shortley convention) carried into this repo's real basis:

    T[m1,m2,m3] = sum_mu U1[m1,mu1] U2[m2,mu2] U3[m3,mu3] (l1 l2 l3; mu1 mu2 mu3),   b_m(r) = sum_mu U_l[m,mu] Y_l^mu(r)

(b = the real basis polynomials of oracle/cg.py, Y = scipy's complex spherical harmonics).  T is real up to one global
phase; after normalisation it must equal the oracle's C up to the documented sign convention of oracle/cg.py
(sign table asserted below), for every (l1, l2, l3) with l <= 2 -- including the odd-sum ones (e.g. the cross product
(1,1,1)) that real Gaunt integrals cannot reach.  Also re-checks the generated header against the oracle."""
import itertools
import os
import re

import numpy as np
import pytest
from sympy.physics.wigner import wigner_3j

from oracle import cg as O

try:  # scipy >= 1.15
    from scipy.special import sph_harm_y

    def _ylm(l, m, theta, phi):
        return sph_harm_y(l, m, theta, phi)
except ImportError:  # older scipy: sph_harm(m, l, azimuth, polar)
    from scipy.special import sph_harm

    def _ylm(l, m, theta, phi):
        return sph_harm(m, l, phi, theta)


def _U(l):
    """[2l+1, 2l+1] complex: real basis polynomial m = sum_mu U[m, mu] Y_l^mu on the unit sphere."""
    rng = np.random.default_rng(7)
    p = rng.normal(size=(200, 3))
    p /= np.linalg.norm(p, axis=1, keepdims=True)
    theta, phi = np.arccos(p[:, 2]), np.arctan2(p[:, 1], p[:, 0])
    Y = np.stack([_ylm(l, mu, theta, phi) for mu in range(-l, l + 1)], 1)   # [P, 2l+1]
    B = O.basis(l, p).astype(complex)                                         # [P, 2l+1]
    U = np.linalg.lstsq(Y, B, rcond=None)[0].T
    assert np.abs(Y @ U.T - B).max() < 1e-12
    return U


def _from_sympy(l1, l2, l3):
    w = np.zeros((2 * l1 + 1, 2 * l2 + 1, 2 * l3 + 1))
    for a, b, c in itertools.product(range(-l1, l1 + 1), range(-l2, l2 + 1), range(-l3, l3 + 1)):
        if a + b + c == 0:
            w[a + l1, b + l2, c + l3] = float(wigner_3j(l1, l2, l3, a, b, c))
    T = np.einsum("ai,bj,ck,ijk->abc", _U(l1), _U(l2), _U(l3), w.astype(complex))
    k = np.argmax(np.abs(T))
    T = T * np.exp(-1j * np.angle(T.reshape(-1)[k]))   # global phase
    assert np.abs(T.imag).max() < 1e-12 * max(1.0, np.abs(T.real).max()), (l1, l2, l3)
    T = T.real
    return T / np.sqrt((T * T).sum())


TRIPLES = [k for k in itertools.product(range(3), repeat=3) if abs(k[0] - k[1]) <= k[2] <= k[0] + k[1]]


@pytest.mark.parametrize("l1,l2,l3", TRIPLES)
def test_oracle_cg_equals_real_basis_wigner_3j(l1, l2, l3):
    C = O.cg(l1, l2, l3)
    T = _from_sympy(l1, l2, l3)
    # T carries an arbitrary overall sign (global phase removal); the oracle fixes it by its own rule -> compare up to sign
    s = np.sign((C * T).sum())
    assert abs(s) == 1
    assert np.abs(C - s * T).max() < 1e-10, (l1, l2, l3, np.abs(C - s * T).max())


def test_l_le_1_constants_are_the_references():
    """l1_tensor_prod.py:91-94: cg000 = 1, cg110 = cg011 = 1/sqrt3 (dot), cg111 = 1/sqrt6 (cross, out = in1 x in2)."""
    assert abs(O.cg(0, 0, 0)[0, 0, 0] - 1) < 1e-14
    assert np.abs(O.cg(1, 1, 0)[:, :, 0] - np.eye(3) / np.sqrt(3)).max() < 1e-12
    assert np.abs(O.cg(0, 1, 1)[0] - np.eye(3) / np.sqrt(3)).max() < 1e-12
    eps = np.zeros((3, 3, 3))
    for i, j, k in ((0, 1, 2), (1, 2, 0), (2, 0, 1)):
        eps[i, j, k], eps[j, i, k] = 1, -1
    assert np.abs(O.cg(1, 1, 1) - eps / np.sqrt(6)).max() < 1e-12


def test_generated_header_matches_oracle():
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scalable-e3-gnn_amd", "csrc",
                        "cg_tables.h")
    text = open(path).read()
    seen = 0
    for m in re.finditer(r"struct CG<(\d),(\d),(\d)> .*? v\[\d\]\[\d\]\[\d\] = (\{.*?\});", text):
        l1, l2, l3 = (int(m.group(i)) for i in (1, 2, 3))
        vals = np.array([float(v) for v in re.findall(r"-?\d+\.\d+(?:e-?\d+)?", m.group(4))])
        C = O.cg(l1, l2, l3)
        assert vals.size == C.size and np.abs(vals.reshape(C.shape) - C).max() < 1e-12, (l1, l2, l3)
        seen += 1
    assert seen == len(TRIPLES)
