"""CPU oracle for the radius graph (ctypes over oracle/radius_graph_oracle.c) + a numpy brute force.
TEST INFRASTRUCTURE ONLY.  Spec: include/e3gnn.h "Radius graph" (builder-defined, parity unpinned
w.r.t. the upstream project — the reference mount holds no graph code)."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "_build", "libgraph_oracle.so")


class RgParams(ctypes.Structure):
    _fields_ = [("lo", ctypes.c_float * 3), ("hi", ctypes.c_float * 3), ("r", ctypes.c_float),
                ("n", ctypes.c_int32 * 3), ("inv", ctypes.c_float * 3), ("bits", ctypes.c_int32)]


def build():
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    src = os.path.join(HERE, "radius_graph_oracle.c")
    if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
        subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-ffp-contract=off", "-o", SO, src, "-lm"], check=True)
    return SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.rg_bruteforce.restype = ctypes.c_int64
        _lib.rg_celllist.restype = ctypes.c_int64
    return _lib


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def params(lo, hi, r):
    p = RgParams()
    for a in range(3):
        p.lo[a], p.hi[a] = float(lo[a]), float(hi[a])
    p.r = float(r)
    lib().rg_grid(ctypes.byref(p))
    return p


def order(pos, p):
    pos = np.ascontiguousarray(pos, dtype=np.float32)
    N = pos.shape[0]
    perm = np.empty(N, np.int32)
    keys = np.empty(N, np.uint32)
    lib().rg_order(_p(pos, ctypes.c_float), ctypes.c_int64(N), ctypes.byref(p), _p(perm, ctypes.c_int32),
                   _p(keys, ctypes.c_uint32))
    return perm, keys


def graph(pos, lo, hi, r, method="celllist"):
    """-> perm [N], rowptr [N+1], src [E]  (new ids)."""
    pos = np.ascontiguousarray(pos, dtype=np.float32)
    N = pos.shape[0]
    p = params(lo, hi, r)
    perm, _ = order(pos, p)
    sp = np.ascontiguousarray(pos[perm])
    rowptr = np.zeros(N + 1, np.int32)
    if method == "bruteforce":
        E = lib().rg_bruteforce(_p(sp, ctypes.c_float), ctypes.c_int64(N), ctypes.c_float(r), _p(rowptr, ctypes.c_int32), None)
        src = np.empty(max(E, 1), np.int32)
        lib().rg_bruteforce(_p(sp, ctypes.c_float), ctypes.c_int64(N), ctypes.c_float(r), _p(rowptr, ctypes.c_int32),
                            _p(src, ctypes.c_int32))
    else:
        E = lib().rg_celllist(_p(sp, ctypes.c_float), ctypes.c_int64(N), ctypes.byref(p), _p(rowptr, ctypes.c_int32), None)
        assert E >= 0
        src = np.empty(max(E, 1), np.int32)
        lib().rg_celllist(_p(sp, ctypes.c_float), ctypes.c_int64(N), ctypes.byref(p), _p(rowptr, ctypes.c_int32),
                          _p(src, ctypes.c_int32))
    return perm, rowptr, src[:E]


def graph_numpy(pos_sorted, r):
    """Independent pure-numpy brute force on already ordered positions (tiny N only)."""
    sp = np.asarray(pos_sorted, np.float32)
    d = sp[:, None, :] - sp[None, :, :]
    d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
    adj = d2 <= np.float32(r) * np.float32(r)
    np.fill_diagonal(adj, False)
    rowptr = np.concatenate([[0], np.cumsum(adj.sum(1))]).astype(np.int32)
    src = np.nonzero(adj)[1].astype(np.int32)
    return rowptr, src
