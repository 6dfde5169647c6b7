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


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("irreps,out,gate,lmax,B", [
    ("32x0e+32x1o+32x2e", "32x0e+32x1o+32x2e", False, 2, 4099),             # update product #2 (l_max = 2)
    ("32x0e+32x1o", "32x0e+32x1o", False, 1, 1000),                         # update product #2 (l_max = 1)
    ("32x0e+32x1o+32x2e", "32x0e+64x0e+32x1o+32x2e", True, 2, 2050),        # gated
])
def test_fused_epilogue_residual_and_scale(irreps, out, gate, lmax, B, dtype):
    """e3_tp_forward_fused_epilogue: `out = product + residual` and the operand scale of `out`, both in the product's
    epilogue, against the product followed by a torch add and ops.pow2_scale."""
    from scalable_e3_gnn_amd.tensor_product import SHTensorProduct
    torch.manual_seed(4)
    mod = SHTensorProduct(irreps, out, lmax).to(DEV).to(dtype)
    g = torch.Generator(device=DEV).manual_seed(5)
    x = (torch.randn(B, mod.in1_dim, device=DEV, generator=g) * 3).to(dtype)
    y = torch.randn(B, mod.in2_dim, device=DEV, generator=g)
    with torch.no_grad():
        plain = mod.forward_fused([(x, None)], y, gate=gate)
        res = (torch.randn(plain.shape, device=DEV, generator=g) * 50).to(dtype)
        res[B // 2, 7] = 7e3                                                          # the maximum sits in one known element
        if dtype == torch.float32:
            got, sc = mod.forward_fused([(x, None)], y, gate=gate, residual=res, out_scale=10)
            want = plain + res
            assert torch.equal(got, want)                                             # the same fp32 add, in the epilogue
            ref = ops.pow2_scale([want], target_log2=10)
            assert float(sc[0]) == float(ref[0]) and float(sc[1]) == float(ref[1])
            assert float(sc[2].view(torch.int32).view(torch.float32)) == float(want.abs().max())
            # scale only
            got2, sc2 = mod.forward_fused([(x, None)], y, gate=gate, out_scale=9)
            assert torch.equal(got2, plain) and float(sc2[0]) == float(ops.pow2_scale([plain], target_log2=9)[0])
        else:
            got = mod.forward_fused([(x, None)], y, gate=gate, residual=res)
            want = plain.float() + res.float()                                        # one rounding of the fp32 sum in the kernel
            assert (got.float() - want).abs().max() <= 2.0 ** -7 * want.abs().max()
