"""Parity of the HIP path (through the C ABI) with the oracle and the reference's golden vectors.

Tolerances (written here once):
  fp64  : 1e-12 relative to the output scale (same arithmetic, different summation order)
  fp32  : 1e-5  relative to the output scale  (BASELINE.json north_star: "within 1e-5 rel fp32")
  bf16  : 2e-2  relative to the output scale  (bf16 storage, fp32 accumulate; the reference rounds
          every intermediate to bf16, the HIP path only the result — documented in DESIGN.md)
"""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_case
from models.segnn.l1_tensor_prod import L1TensorProduct
from oracle import l1tp_oracle as O
from scalable_e3_gnn_amd import Irreps

pytestmark = pytest.mark.gpu
META = json.load(open(os.path.join(GOLDEN, "l1tp_meta.json")))
TOL = {"float64": 1e-12, "float32": 1e-5, "bfloat16": 2e-2}
TDT = {"float64": torch.float64, "float32": torch.float32, "bfloat16": torch.bfloat16}
DEV = "cuda:0"


def close(got, ref, tol, what=""):
    got = got.detach().double().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    if ref.size == 0:
        return
    scale = max(float(np.abs(ref).max()), 1e-30)
    err = float(np.abs(got - ref).max()) / scale
    assert err <= tol, f"{what}: rel-to-scale error {err:.3e} > {tol:.1e}"


def module_from_case(c, z, kernel=0):
    torch.manual_seed(c["seed"])
    mod = L1TensorProduct(Irreps(c["in1"]), Irreps(c["out"]) if c["out"] else None, **c["kwargs"])
    mod = mod.to(TDT[c["dtype"]])
    sd = {k[3:]: torch.tensor(z[k]).to(TDT[c["dtype"]]) for k in z.files if k.startswith("sd_")}
    mod.load_state_dict(sd, strict=True)
    mod.kernel = kernel
    return mod.to(DEV)


@pytest.mark.parametrize("kernel", [1, 0])
@pytest.mark.parametrize("name", sorted(META["cases"]))
def test_forward_matches_reference_golden(name, kernel):
    c = META["cases"][name]
    z = load_case(name)
    mod = module_from_case(c, z, kernel)
    dt = TDT[c["dtype"]]
    x = torch.tensor(z["in1"]).to(dt).to(DEV)
    y = torch.tensor(z["in2"]).to(dt).to(DEV)
    out = mod(x, y)
    assert out.dtype == dt and out.is_contiguous() and out.shape == z["out"].shape
    close(out, z["out"], TOL[c["dtype"]], name)


@pytest.mark.parametrize("path", ["kernels", "gemm", "dual"])
@pytest.mark.parametrize("name", sorted(META["cases"]))
def test_backward_matches_reference_golden(name, path, monkeypatch):
    """path: the operator's own 16-row kernels (small B) / operands -> library GEMMs -> contract on the general plan
    (`tensor_product.tp_backward`, what large B runs) / `dual`: in2 needs no gradient (the training case), grad_in1 = the
    forward kernel on the transposed operator (`L1TensorProduct._dual_forward`, fp32), grad_W = operands + GEMM."""
    from scalable_e3_gnn_amd import tensor_product as TPM
    monkeypatch.setattr(TPM, "_BWD_GEMM_MIN_ROWS", 0 if path != "kernels" else 1 << 30)
    c = META["cases"][name]
    z = load_case(name)
    mod = module_from_case(c, z)
    dt = TDT[c["dtype"]]
    x = torch.tensor(z["in1"]).to(dt).to(DEV).requires_grad_(True)
    y = torch.tensor(z["in2"]).to(dt).to(DEV).requires_grad_(path != "dual")
    go = torch.tensor(z["grad_out"]).to(dt).to(DEV)
    out = mod(x, y)
    (out * go).sum().backward()
    tol = TOL[c["dtype"]] * (4 if c["dtype"] != "float64" else 1)
    close(x.grad, z["grad_in1"], tol, name + " grad_in1")
    if path != "dual":
        close(y.grad, z["grad_in2"], tol, name + " grad_in2")
    for k, p in mod.named_parameters():
        close(p.grad, z["grad_" + k], tol, f"{name} grad_{k}")


@pytest.mark.parametrize("kernel", [1, 0])
@pytest.mark.parametrize("irreps,out,B", [
    ("32x0e+32x1o", None, 4099),
    ("8x0e+8x1o", None, 1000),
    ("64x0e+64x1o", None, 515),
    ("16x0e+16x1o+16x0e+16x1o+1x0e", "32x0e+16x1o", 777),
    ("5x0e+3x0o+7x1o+2x1e", "9x0e+1x0o+4x1e+6x1o", 333),
    ("33x0e+31x1o", "35x0e+30x1o", 129),
])
def test_forward_vs_oracle_seeded(irreps, out, B, kernel):
    """fp32 HIP vs fp64 oracle on seeded inputs (sizes the oracle finishes in seconds)."""
    torch.manual_seed(0)
    mod = L1TensorProduct(Irreps(irreps), Irreps(out) if out else None).to(DEV)
    mod.kernel = kernel
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, mod.in1_dim, generator=g)
    y = torch.randn(B, 4, generator=g)
    lay = O.make_layout(irreps, out)
    W = {c: getattr(mod, "weights_" + c).detach().cpu().numpy() for c in O.CLASSES if hasattr(mod, "weights_" + c)}
    N = {c: getattr(mod, "norm_" + c).cpu().numpy() for c in O.CLASSES}
    ref = O.forward_closed_form(lay, x.numpy(), y.numpy(), W, N)
    got = mod(x.to(DEV), y.to(DEV))
    close(got, ref, TOL["float32"], f"{irreps} k{kernel}")
    # the reference-pattern fp32 CPU restatement is no closer to fp64 truth than the HIP path is allowed to be
    ff = O.forward_faithful(lay, x, y, {k: torch.tensor(v) for k, v in W.items()}, {k: torch.tensor(v) for k, v in N.items()})
    close(got, ff.numpy(), TOL["float32"], f"{irreps} k{kernel} vs faithful fp32")


def test_generic_and_mfma_agree_and_edge_shapes():
    torch.manual_seed(5)
    mod = L1TensorProduct(Irreps("32x0e+32x1o")).to(DEV)
    for B in (1, 31, 32, 33, 64, 255, 257):
        x = torch.randn(B, 128, device=DEV)
        y = torch.randn(B, 4, device=DEV)
        mod.kernel = 1
        a = mod(x, y)
        mod.kernel = 0
        b = mod(x, y)
        close(b, a.detach().double().cpu().numpy(), 1e-5, f"B={B}")
    # empty batch
    out = mod(torch.zeros(0, 128, device=DEV), torch.zeros(0, 4, device=DEV))
    assert out.shape == (0, 128)
    # in2 broadcast == expanded
    x = torch.randn(100, 128, device=DEV)
    y1 = torch.randn(1, 4, device=DEV)
    close(mod(x, y1), mod(x, y1.expand(100, 4).contiguous()).detach().double().cpu().numpy(), 1e-6, "broadcast")
    # row-strided (non-contiguous) inputs
    xx = torch.randn(100, 200, device=DEV)
    close(mod(xx[:, 5:133], y1), mod(xx[:, 5:133].contiguous(), y1).detach().double().cpu().numpy(), 1e-7, "strided")


def test_linearity_equivariance_parity_at_scale():
    """Size-independent properties on a batch far larger than any fixture (2^20 rows)."""
    torch.manual_seed(7)
    irreps = "32x0e+32x1o"
    mod = L1TensorProduct(Irreps(irreps)).to(DEV)
    B = 1 << 20
    g = torch.Generator(device=DEV).manual_seed(11)
    x1 = torch.randn(B, 128, device=DEV, generator=g)
    x2 = torch.randn(B, 128, device=DEV, generator=g)
    y = torch.randn(B, 4, device=DEV, generator=g)
    o1, o2 = mod(x1, y), mod(x2, y)
    o12 = mod(2.0 * x1 - 0.5 * x2, y)
    lin = 2.0 * o1 - 0.5 * o2
    assert ((o12 - lin).abs().max() / lin.abs().max()).item() < 2e-6
    # rotation: rotate every 1o block and Y1 by R -> outputs' 1o blocks rotate by R, scalars invariant
    q, _ = torch.linalg.qr(torch.randn(3, 3, dtype=torch.float64))
    R = (q * torch.sign(torch.linalg.det(q))).float().to(DEV)

    def rot(v):
        v = v.clone()
        v[:, 32:] = (v[:, 32:].reshape(B, 32, 3) @ R.T).reshape(B, 96)
        return v
    yr = y.clone()
    yr[:, 1:] = y[:, 1:] @ R.T
    orot = mod(rot(x1), yr)
    assert ((orot - rot(o1)).abs().max() / o1.abs().max()).item() < 5e-6
    # inversion: 1o and Y1 flip sign -> 0e invariant, 1o flips
    xi = x1.clone(); xi[:, 32:] *= -1
    yi = y.clone(); yi[:, 1:] *= -1
    oi = mod(xi, yi)
    want = o1.clone(); want[:, 32:] *= -1
    assert ((oi - want).abs().max() / o1.abs().max()).item() < 1e-6


def test_gradcheck_fp64_small():
    torch.manual_seed(3)
    mod = L1TensorProduct(Irreps("2x0e+1x0o+2x1o+1x1e"), Irreps("2x0e+1x0o+1x1e+2x1o")).double().to(DEV)
    x = torch.randn(3, mod.in1_dim, dtype=torch.float64, device=DEV, requires_grad=True)
    y = torch.randn(3, 4, dtype=torch.float64, device=DEV, requires_grad=True)
    assert torch.autograd.gradcheck(lambda a, b: mod(a, b), (x, y), eps=1e-6, atol=1e-7, rtol=1e-6)
    yb = torch.randn(1, 4, dtype=torch.float64, device=DEV, requires_grad=True)
    assert torch.autograd.gradcheck(lambda a, b: mod(a, b), (x, yb), eps=1e-6, atol=1e-7, rtol=1e-6)


def test_weight_update_invalidates_packed_cache():
    torch.manual_seed(1)
    mod = L1TensorProduct(Irreps("8x0e+8x1o")).to(DEV)
    x = torch.randn(10, 32, device=DEV)
    y = torch.randn(10, 4, device=DEV)
    a = mod(x, y)
    with torch.no_grad():
        mod.weights_l0e.mul_(2.0)
    b = mod(x, y)
    assert torch.allclose(b[:, :8], 2 * a[:, :8], rtol=1e-6, atol=1e-6) and torch.allclose(b[:, 8:], a[:, 8:])
