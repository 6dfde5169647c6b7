"""The C-ABI library loads, exports every symbol declared in include/*.h, and its host-only entry points
(plan bookkeeping) agree with the oracle.  No compute calls: runs without a GPU."""
import ctypes
import os
import re

import pytest

from conftest import REPO
from oracle import l1tp_oracle as O
from scalable_e3_gnn_amd import _lib


def declared_symbols():
    names = set()
    for fn in os.listdir(os.path.join(REPO, "include")):
        if fn.endswith(".h"):
            text = open(os.path.join(REPO, "include", fn)).read()
            text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
            names |= set(re.findall(r"\b(e3_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    syms = declared_symbols()
    assert len(syms) >= 12
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, f"declared in include/*.h but not exported: {missing}"
    assert set(_lib.SIGNATURES) <= set(syms), "python binding lists an undeclared symbol"
    assert set(syms) <= set(_lib.SIGNATURES), f"header symbols without a binding: {set(syms) - set(_lib.SIGNATURES)}"
    assert lib.e3_abi_version() == 3
    assert lib.e3_status_string(0) == b"ok"


@pytest.mark.parametrize("in1,out", [
    ("8x0e+8x1o", "8x0e+8x1o"),
    ("3x0e+2x0o+4x1o+5x1e", "6x0e+2x0o+3x1e+7x1o"),
    ("2x1o+3x0e+1x1e+2x0e+2x1o+1x0o", "1x1e+2x0e+2x1o+1x0o+3x1o+2x0e"),
    ("3x1e", "2x0o+2x1e+3x1o"),
])
def test_plan_bookkeeping_matches_oracle(in1, out):
    lib = _lib.load()
    a, na = _lib.blocks_array(O.parse_blocks(in1))
    b, nb = _lib.blocks_array(O.parse_blocks(out))
    h = ctypes.c_void_p()
    assert lib.e3_l1tp_plan_create(a, na, b, nb, ctypes.byref(h)) == 0
    lay = O.make_layout(in1, out)
    assert lib.e3_l1tp_in1_dim(h) == lay.in1_dim and lib.e3_l1tp_out_dim(h) == lay.out_dim
    for ci, cls in enumerate(O.CLASSES):
        r, c = ctypes.c_int(), ctypes.c_int()
        assert lib.e3_l1tp_weight_shape(h, ci, ctypes.byref(r), ctypes.byref(c)) == 0
        assert (r.value, c.value) == (lay.wshape[cls] or (0, 0))
        assert lib.e3_l1tp_norm_len(h, ci) == len(lay.o[cls])
    assert lib.e3_l1tp_packed_bytes(h, 0) > 0
    assert lib.e3_l1tp_plan_destroy(h) == 0


def test_plan_rejects_bad_irreps():
    lib = _lib.load()
    h = ctypes.c_void_p()
    for blocks in ([(0, 1, 4)], [(0, 1, 4), (2, 1, 1)], [(1, 0, 1)], [(1, -1, -1)]):
        a, na = _lib.blocks_array(blocks)
        good, ng = _lib.blocks_array([(0, 1, 1), (1, -1, 1)])
        assert lib.e3_l1tp_plan_create(a, na, good, ng, ctypes.byref(h)) == 2  # E3_ERR_BAD_IRREPS
        assert lib.e3_l1tp_plan_create(good, ng, a, na, ctypes.byref(h)) == 2
    assert lib.e3_l1tp_plan_create(None, 0, None, 0, ctypes.byref(h)) == 1
