"""Operand scales (e3_pow2_scale / e3_add_pow2_scale, csrc/e3_scale.hip): the power of two that puts max |x| of the listed
segments at 2^target -- dense float4 path, strided float4 path (column blocks of a wider tensor) and the scalar path
(odd widths / unaligned views), against torch."""
import math

import pytest
import torch

import models  # noqa: F401
from scalable_e3_gnn_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("target", [10, 9])
def test_pow2_scale_paths(target):
    g = torch.Generator(device=DEV).manual_seed(3)
    wide = torch.randn(5000, 300, device=DEV, generator=g) * 37.0
    cases = {
        "dense": [torch.randn(7001, 288, device=DEV, generator=g) * 3.0],
        "strided float4": [wide[:, 8:296]],                 # ld 300, cols 288, 16-byte aligned start
        "scalar (odd start)": [wide[:, 3:290]],             # unaligned, cols 287
        "column": [wide[:, 5]],                             # 1-d strided view
        "two segments": [torch.randn(4000, 288, device=DEV, generator=g), wide[:, 0:128] * 0.01],
        "zeros": [torch.zeros(64, 32, device=DEV)],
    }
    for name, ts in cases.items():
        ts2 = [t if t.dim() == 2 else t.unsqueeze(1) for t in ts]
        sc = ops.pow2_scale(ts2, target_log2=target)
        s, inv = float(sc[0]), float(sc[1])
        assert s * inv == 1.0, name
        m = max(float(t.abs().max()) for t in ts2)
        if m == 0:
            assert s == 1.0, name
            continue
        # s is the power of two with 2^target <= s * max < 2^(target+1)  (pow2_scale_from_bits, e3_tp_mfma_core.h)
        assert math.log2(s) == round(math.log2(s)), name
        assert 2.0 ** target <= s * m < 2.0 ** (target + 1), (name, s, m)
    # a planted maximum is found wherever it sits (first / last row, last column)
    t = torch.zeros(3001, 288, device=DEV)
    for r, c in [(0, 0), (3000, 287), (1500, 3)]:
        t.zero_()
        t[r, c] = -1234.5
        s = float(ops.pow2_scale([t])[0])
        assert 1024.0 <= s * 1234.5 < 2048.0, (r, c, s)


def test_add_pow2_scale_matches_separate_ops():
    g = torch.Generator(device=DEV).manual_seed(4)
    h = torch.randn(9000, 288, device=DEV, generator=g) * 11.0
    u = torch.randn(9000, 288, device=DEV, generator=g)
    out, sc = ops.add_pow2_scale(h, u)
    assert torch.equal(out, h + u)
    assert torch.equal(sc[:2], ops.pow2_scale([h + u])[:2])
