"""Pins the CPU oracle (oracle/l1tp_oracle.py) against
  * golden vectors captured from the unmodified reference (tests/golden, make_golden.py), and
  * the known-answer material recorded in SURVEY.md §4.
fp64 bar: 1e-12; fp32 bar: 1e-6 relative to the output scale (BLAS order differs)."""
import json
import math

import numpy as np
import pytest
import torch

from conftest import load_case
from oracle import l1tp_oracle as O


def case_names(meta):
    return sorted(meta["cases"].keys())


def _weights_norms(z):
    W = {k: z["sd_weights_" + k] for k in O.CLASSES if "sd_weights_" + k in z}
    N = {k: z["sd_norm_" + k] for k in O.CLASSES}
    return W, N


def test_all_cases_present(golden_meta):
    assert len(golden_meta["cases"]) >= 19


@pytest.mark.parametrize("name", json.load(open(__file__.replace("test_oracle_golden.py", "golden/l1tp_meta.json")))["cases"].keys())
def test_layout_norms_forward(name, golden_meta):
    c = golden_meta["cases"][name]
    z = load_case(name)
    lay = O.make_layout(c["in1"], c["out"])
    nr = O.normalisation(c["in1"], c["out"], **c["kwargs"])
    # a-1 masks
    for cls in O.CLASSES:
        assert np.flatnonzero(np.array(c["masks"]["iri1_" + cls])).tolist() == lay.i1[cls].tolist()
        assert np.flatnonzero(np.array(c["masks"]["iro_" + cls])).tolist() == lay.o[cls].tolist()
    assert lay.in1_dim == c["attrs"]["in1_dim"]
    # a-2 parameter presence / shapes
    for cls in O.CLASSES:
        key = "sd_weights_" + cls
        assert (lay.wshape[cls] is None) == (key not in z)
        if lay.wshape[cls] is not None:
            assert tuple(z[key].shape) == lay.wshape[cls]
    # a-4 norms (bit-equal fp32 buffers), instructions, path weights
    for cls in O.CLASSES:
        assert np.array_equal(nr.norms[cls], z["init_norm_" + cls])
    assert [list(i[:5]) for i in nr.instructions] == [i[:5] for i in c["instructions"]]
    assert [i[5] for i in nr.instructions] == [i[5] for i in c["instructions"]]
    assert [list(i[6]) for i in nr.instructions] == [i[6] for i in c["instructions"]]
    # a-5..a-9 forward
    W, N = _weights_norms(z)
    ref = z["out"]
    out64 = O.forward_closed_form(lay, z["in1"], z["in2"], W, N)
    scale = max(1.0, float(np.abs(ref).max())) if ref.size else 1.0
    tol = {"float64": 1e-12, "float32": 2e-6, "bfloat16": 2e-2}[c["dtype"]]
    assert out64.shape == ref.shape
    if ref.size:
        assert np.abs(out64 - ref).max() <= tol * scale
    if c["dtype"] != "bfloat16":
        tdt = getattr(torch, c["dtype"])
        f = O.forward_faithful(lay, torch.tensor(z["in1"], dtype=tdt), torch.tensor(z["in2"], dtype=tdt),
                               {k: torch.tensor(v, dtype=tdt) for k, v in W.items()},
                               {k: torch.tensor(v, dtype=tdt) for k, v in N.items()})
        assert f.is_contiguous() and f.dtype == tdt
        if ref.size:
            assert np.abs(f.double().numpy() - ref).max() <= tol * scale


def test_survey_kat_deterministic(golden_meta):
    """SURVEY.md §4 'Deterministic KAT (no RNG)' — numbers typed from the survey, not from the fixture."""
    row0 = [0.46823153890958696, 0.08049382178673467, 0.10206207633018494, 0.7696067673734669,
            0.27274755952209656, 0.27209708202841815, 0.03952847186917966, 0.042919478655671875,
            1.04575909018017, 0.3238325847431225, -0.49013308679388995, -0.15133187390373418]
    row1 = [-0.07890732226839049, -0.13046719360626435, 0.3061862289905548, -0.6619368792237618,
            0.458351925822707, -0.03915775952234179, 0.006491111180802693, -0.13127673557696248,
            -0.48364197561308725, 0.4966241979746926, 0.019473333542407992, 0.3159368725612555]
    lay = O.make_layout("2x0e+1x0o+2x1o+1x1e", "2x0e+1x0o+1x1e+2x1o")
    nr = O.normalisation("2x0e+1x0o+2x1o+1x1e", "2x0e+1x0o+1x1e+2x1o")
    W = {}
    for cls, shp in lay.wshape.items():
        i = np.arange(shp[0])[:, None]
        j = np.arange(shp[1])[None, :]
        W[cls] = (((7 * i + 3 * j) % 5) - 2) / 4
    e = np.arange(2)[:, None]
    d = np.arange(12)[None, :]
    x = (((12 * e + d) % 7) - 3) / 2
    y = np.array([[1, .5, -1, 2], [1, -1.5, .25, .75]])
    out = O.forward_closed_form(lay, x, y, W, nr.norms)  # norms are fp32-rounded (Q4)
    assert np.abs(out - np.array([row0, row1])).max() < 1e-14
    assert np.abs(out - np.array(golden_meta["kat"]["out"])).max() < 1e-14
    assert nr.norms["l0e"][0] == np.float32(0.40824830532073975)
    assert nr.norms["l1e"][0] == np.float32(0.8660253882408142)
    assert nr.norms["l1o"][0] == np.float32(0.7745966911315918)


def test_survey_norm_tables():
    """SURVEY.md §4 norm table (Q1: parity-blind count gives sqrt(1/14) twice)."""
    nr = O.normalisation("8x0e+8x1o")
    assert nr.norms["l0e"][0] == np.float32(0.25)
    assert abs(float(nr.norms["l1o"][0]) - math.sqrt(3 / 16)) < 1e-7
    assert [i[:3] for i in nr.instructions] == [(0, 0, 0), (1, 1, 0), (1, 0, 1), (0, 1, 1)]
    assert nr.instructions[0][6] == (8, 1, 8)
    nr = O.normalisation("3x0e+2x0o+4x1o+5x1e", "6x0e+2x0o+3x1e+7x1o")
    got = [float(nr.norms[c][0]) for c in ("l0e", "l0o", "l1e", "l1o")]
    want = [math.sqrt(1 / 14), math.sqrt(1 / 14), math.sqrt(3 / 11), math.sqrt(3 / 12)]
    assert np.allclose(got, want, rtol=0, atol=1e-7)
    lay = O.make_layout("3x0e+2x0o+4x1o+5x1e", "6x0e+2x0o+3x1e+7x1o")
    assert lay.wshape == {"l0e": (7, 6), "l0o": (7, 2), "l1e": (11, 3), "l1o": (12, 7)}
    assert {c: len(nr.norms[c]) for c in O.CLASSES} == {"l0e": 6, "l0o": 2, "l1e": 9, "l1o": 21}


def test_error_paths_match_reference(golden_meta):
    errs = golden_meta["errors"]
    for name in ("scalar_only_in", "scalar_only_out"):
        assert errs[name]["raised"] == "AssertionError"
        with pytest.raises(AssertionError):
            O.make_layout(errs[name]["spec"]["in1"], errs[name]["spec"].get("out"))
    for name in ("bad_in1_var_len", "bad_in2_var_len", "bad_out_var_len", "norm_norm", "norm_path"):
        with pytest.raises(AssertionError) as ei:
            O.normalisation(errs[name]["spec"]["in1"], None, **errs[name]["spec"]["kwargs"])
        assert str(ei.value) == errs[name]["message"]
    assert O.normalisation("4x0e+4x1o", None, "none", "none").is_norm is False  # Q2


def test_equivariance_of_oracle():
    """SO(3) equivariance + parity of the restated formulas (the reference has no such test)."""
    rng = np.random.default_rng(0)
    lay = O.make_layout("3x0e+2x0o+4x1o+5x1e", "6x0e+2x0o+3x1e+7x1o")
    nr = O.normalisation("3x0e+2x0o+4x1o+5x1e", "6x0e+2x0o+3x1e+7x1o")
    W = {c: rng.uniform(-1, 1, s) for c, s in lay.wshape.items()}
    B = 5
    x = rng.normal(size=(B, lay.in1_dim))
    y = rng.normal(size=(B, 4))
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    R = q * np.sign(np.linalg.det(q))

    def rot(v, blocks, R, inv):
        v = v.copy()
        col = 0
        for l, p, mul in blocks:
            if l == 1:
                blk = v[:, col:col + 3 * mul].reshape(-1, mul, 3) @ R.T
                if inv and p == -1:
                    blk = -blk
                v[:, col:col + 3 * mul] = blk.reshape(-1, 3 * mul)
            elif inv and p == -1:
                v[:, col:col + mul] *= -1
            col += (2 * l + 1) * mul
        return v

    for inv in (False, True):
        o = O.forward_closed_form(lay, x, y, W, nr.norms)
        xr = rot(x, lay.in1_blocks, R, inv)
        yr = rot(y, O.SH_BLOCKS, R, inv)
        o2 = O.forward_closed_form(lay, xr, yr, W, nr.norms)
        assert np.abs(o2 - rot(o, lay.out_blocks, R, inv)).max() < 1e-12
