"""e3_tp_backward (HIP) against torch autograd over the oracle's torch-CPU statement of the same contraction."""
import numpy as np
import pytest
import torch

import models  # noqa: F401
from oracle import tp_oracle as T
from scalable_e3_gnn_amd.tensor_product import SHTensorProduct

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def oracle_grads(mod, in1_irreps, out_irreps, lmax, x, y, gout, dtype):
    W = {c: getattr(mod, "weights_" + c).detach().cpu().to(dtype).requires_grad_(True) for c in T.CLASSES
         if hasattr(mod, "weights_" + c)}
    N = {c: getattr(mod, "norm_" + c).detach().cpu().to(dtype) for c in T.CLASSES}
    xc, yc = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
    out = T.forward_torch_cpu(in1_irreps, out_irreps, lmax, xc, yc if yc.shape[0] == xc.shape[0] else yc.expand(xc.shape[0], -1), W, N)
    out.backward(gout)
    return out.detach(), xc.grad, yc.grad, {c: W[c].grad for c in W}


@pytest.mark.parametrize("in1,out,lmax,dtype,B,bcast", [
    ("3x0e+2x0o+4x1o+5x1e+2x2e+3x2o", "6x0e+2x0o+3x1e+7x1o+2x2o+3x2e", 2, "float64", 37, False),
    ("8x0e+8x1o+8x2e", "8x0e+16x0e+8x1o+8x2e", 2, "float32", 131, False),
    ("8x0e+8x1o", "8x0e+8x1o", 1, "float64", 19, True),
    ("32x0e+32x1o+32x2e", "32x0e+64x0e+32x1o+32x2e", 2, "float32", 70, False),   # forward on the MFMA kernel
])
@pytest.mark.parametrize("path", ["generic", "gemm", "gemm_chunked"])
def test_backward_vs_oracle_autograd(in1, out, lmax, dtype, B, bcast, path, monkeypatch):
    """Both backward paths of `tensor_product.tp_backward`: the two generic kernels (small B) and operands -> GEMMs ->
    contract (large B; `gemm_chunked` shrinks the workspace so that the row loop takes several passes)."""
    from scalable_e3_gnn_amd import tensor_product as TPM
    monkeypatch.setattr(TPM, "_BWD_GEMM_MIN_ROWS", 1 << 30 if path == "generic" else 0)
    if path == "gemm_chunked":
        monkeypatch.setattr(TPM, "_BWD_WORKSPACE_BYTES", 1 << 16)
        B = B * 9
    dt = getattr(torch, dtype)
    torch.manual_seed(0)
    mod = SHTensorProduct(in1, out, lmax).to(dt).to(DEV)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, mod.in1_dim, generator=g, dtype=dt)
    y = torch.randn(1 if bcast else B, mod.in2_dim, generator=g, dtype=dt)
    gout = torch.randn(B, mod.out_dim, generator=g, dtype=dt)
    want_out, want_gx, want_gy, want_gw = oracle_grads(mod, in1, out, lmax, x, y, gout, dt)
    xd, yd = x.to(DEV).requires_grad_(True), y.to(DEV).requires_grad_(True)
    got = mod(xd, yd)
    got.backward(gout.to(DEV))
    tol = 1e-11 if dtype == "float64" else 2e-5

    def rel(a, b):
        return float((a.double().cpu() - b.double()).abs().max() / b.double().abs().max())
    assert rel(got.detach(), want_out) < (1e-12 if dtype == "float64" else 1e-5)
    assert rel(xd.grad, want_gx) < tol
    assert yd.grad.shape == y.shape and rel(yd.grad, want_gy) < tol
    for c, gw in want_gw.items():
        assert rel(getattr(mod, "weights_" + c).grad, gw) < tol, c


def test_backward_partial_requests_and_empty_batch():
    torch.manual_seed(2)
    mod = SHTensorProduct("4x0e+4x1o", "4x0e+4x1o", 1).double().to(DEV)
    x = torch.randn(9, mod.in1_dim, dtype=torch.float64, device=DEV)
    y = torch.randn(9, mod.in2_dim, dtype=torch.float64, device=DEV)
    # only the weights ask for gradients
    out = mod(x, y)
    out.sum().backward()
    assert all(getattr(mod, "weights_" + c).grad is not None for c in T.CLASSES if hasattr(mod, "weights_" + c))
    # only in1
    for p in mod.parameters():
        p.requires_grad_(False)
    xr = x.clone().requires_grad_(True)
    mod(xr, y).sum().backward()
    assert xr.grad is not None and torch.isfinite(xr.grad).all()
    # empty batch
    xe = torch.empty(0, mod.in1_dim, dtype=torch.float64, device=DEV, requires_grad=True)
    ye = torch.empty(0, mod.in2_dim, dtype=torch.float64, device=DEV)
    mod(xe, ye).sum().backward()
    assert xe.grad.shape == xe.shape


@pytest.mark.parametrize("in1,out,lmax,B", [
    ("32x0e+32x1o+32x0e+32x1o+1x0e", "32x0e+32x0e+32x1o", 1, 5000),                      # message product #1, l_max = 1
    ("32x0e+32x1o+32x2e+32x0e+32x1o+32x2e+1x0e", "32x0e+64x0e+32x1o+32x2e", 2, 3001),    # message product #1, l_max = 2
    ("5x0e+3x0o+7x1o+2x1e", "9x0e+1x0o+4x1e+6x1o", 1, 777),                              # every class, odd sizes
    ("8x0e+8x1o+8x2e", "8x0e+16x0e+8x1o+8x2e", 2, 33),                                   # fewer rows than one row tile
])
def test_fused_weight_gradient_vs_gemm_path(in1, out, lmax, B, monkeypatch):
    """e3_tp_backward_weights (features in LDS, fp32 MFMA) against the operand pass + GEMM path and the fp64 kernels."""
    from scalable_e3_gnn_amd import tensor_product as TPM
    torch.manual_seed(0)
    mod = SHTensorProduct(in1, out, lmax).to(DEV)
    mod64 = SHTensorProduct(in1, out, lmax).double().to(DEV)
    mod64.load_state_dict({k: v.double() for k, v in mod.state_dict().items()})
    g = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn(B, mod.in1_dim, device=DEV, generator=g)
    y = torch.randn(B, mod.in2_dim, device=DEV, generator=g)
    go = torch.randn(B, mod.out_dim, device=DEV, generator=g)
    monkeypatch.setattr(TPM, "_BWD_GEMM_MIN_ROWS", 0)
    grads = {}
    for fused in (True, False):
        monkeypatch.setattr(TPM, "_BWD_FUSED_WGRAD", fused)
        for p in mod.parameters():
            p.grad = None
        mod(x, y).backward(go)
        grads[fused] = {k: p.grad.clone() for k, p in mod.named_parameters()}
    mod64(x.double(), y.double()).backward(go.double())
    for k, p in mod64.named_parameters():
        ref = p.grad
        for fused in (True, False):
            err = float((grads[fused][k].double() - ref).abs().max() / ref.abs().max())
            assert err < 2e-5, (k, fused, err)
