"""General SH tensor product (l <= 2): CG tables, reduction to the reference-pinned L1TP, GPU parity
with the numpy oracle, equivariance.  Builder-defined (parity unpinned w.r.t. upstream)."""
import numpy as np
import pytest
import torch

from oracle import cg as CG
from oracle import l1tp_oracle as O1
from oracle import tp_oracle as T


def test_cg_tables_match_reference_constants_and_are_invariant():
    assert CG.cg(0, 0, 0)[0, 0, 0] == pytest.approx(1.0)
    assert np.allclose(CG.cg(1, 1, 0)[:, :, 0], np.eye(3) * O1.C3, atol=1e-14)
    assert np.allclose(CG.cg(0, 1, 1)[0], np.eye(3) * O1.C3, atol=1e-14)
    eps = np.zeros((3, 3, 3))
    for i, j, k in [(0, 1, 2), (1, 2, 0), (2, 0, 1)]:
        eps[i, j, k], eps[j, i, k] = 1, -1
    assert np.allclose(CG.cg(1, 1, 1), eps * O1.C6, atol=1e-14)
    rng = np.random.default_rng(5)
    R = CG.random_rotation(rng)
    for (l1, l2, l3), C in CG.all_tables().items():
        D1, D2, D3 = (CG.rotation_matrices(l, R) for l in (l1, l2, l3))
        assert np.allclose(np.einsum("ai,bj,ck,ijk->abc", D1, D2, D3, C), C, atol=1e-12)
        assert abs((C * C).sum() - 1) < 1e-12
    assert CG.cg(0, 1, 2) is None and len(CG.all_tables()) == 15


def test_generated_header_matches_oracle_tables():
    import os, re
    from conftest import REPO
    text = open(os.path.join(REPO, "scalable-e3-gnn_amd", "csrc", "cg_tables.h")).read()
    for (l1, l2, l3), C in CG.all_tables().items():
        m = re.search(rf"CG<{l1},{l2},{l3}> .*?= (\{{.*?\}});", text)
        vals = np.array([float(v) for v in re.findall(r"-?\d+\.\d+(?:e-?\d+)?", m.group(1))])
        assert np.array_equal(vals, C.reshape(-1)), (l1, l2, l3)


def test_oracle_reduces_to_l1tp_oracle():
    rng = np.random.default_rng(0)
    for in1, out in (("8x0e+8x1o", "8x0e+8x1o"), ("3x0e+2x0o+4x1o+5x1e", "6x0e+2x0o+3x1e+7x1o"),
                     ("2x1o+3x0e+1x1e+2x0e+2x1o+1x0o", "1x1e+2x0e+2x1o+1x0o+3x1o+2x0e")):
        lay = O1.make_layout(in1, out)
        sh = T.shapes(in1, out, 1)
        W = {}
        for c in O1.CLASSES:
            assert (lay.wshape[c] or None) == (sh[c] or None)
            if lay.wshape[c]:
                W[c] = rng.uniform(-1, 1, lay.wshape[c])
        x, y = rng.normal(size=(7, lay.in1_dim)), rng.normal(size=(7, 4))
        norms = {c: rng.uniform(0.5, 1.5, len(lay.o[c])) for c in O1.CLASSES}
        a = O1.forward_closed_form(lay, x, y, W, norms)
        b = T.forward(in1, out, 1, x, y, W, {**{c: np.zeros(0) for c in T.CLASSES}, **norms})
        assert np.abs(a - b).max() < 1e-13


@pytest.mark.gpu
@pytest.mark.parametrize("in1,out,lmax_sh,dtype", [
    ("8x0e+8x1o", "8x0e+8x1o", 1, "float32"),
    ("8x0e+8x1o+8x2e", "8x0e+8x1o+8x2e", 2, "float64"),
    ("8x0e+8x1o+8x2e", "8x0e+8x1o+8x2e", 2, "float32"),
    ("3x0e+2x0o+4x1o+5x1e+2x2e+3x2o", "6x0e+2x0o+3x1e+7x1o+2x2o+3x2e", 2, "float64"),
    ("32x0e+32x1o+32x2e+32x0e+32x1o+32x2e+1x0e", "96x0e+32x1o+32x2e", 2, "float32"),
    ("32x0e+32x1o+32x2e", "1x1o", 2, "float32"),            # readout shape: l=2 inputs, one l=1 output channel
    ("1x0e+1x1o", "32x0e+32x1o+32x2e", 2, "float32"),       # embedding shape: single-channel input chunks
    ("32x0e+32x1o", "1x1o", 1, "float32"),
])
def test_gpu_forward_vs_oracle(in1, out, lmax_sh, dtype):
    from scalable_e3_gnn_amd.tensor_product import SHTensorProduct
    torch.manual_seed(0)
    dt = getattr(torch, dtype)
    mod = SHTensorProduct(in1, out, lmax_sh).to(dt).to("cuda:0")
    B = 203
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, mod.in1_dim, generator=g, dtype=torch.float64)
    y = torch.randn(B, mod.in2_dim, generator=g, dtype=torch.float64)
    W = {c: getattr(mod, "weights_" + c).detach().double().cpu().numpy() for c in T.CLASSES if hasattr(mod, "weights_" + c)}
    N = {c: getattr(mod, "norm_" + c).double().cpu().numpy() for c in T.CLASSES}
    # default norms follow sqrt((2l+1)/fan_in)
    dn = T.default_norms(in1, out, lmax_sh)
    for c in T.CLASSES:
        assert np.allclose(N[c], dn[c], rtol=1e-6)
    want = T.forward(in1, out, lmax_sh, x.numpy(), y.numpy(), W, N)
    with torch.no_grad():
        got = mod(x.to(dt).to("cuda:0"), y.to(dt).to("cuda:0")).double().cpu().numpy()
    err = np.abs(got - want).max() / np.abs(want).max()
    assert err < (1e-12 if dtype == "float64" else 1e-5), err


@pytest.mark.gpu
def test_gpu_reduces_to_l1tp_kernel():
    from models.segnn.l1_tensor_prod import L1TensorProduct
    from scalable_e3_gnn_amd.tensor_product import SHTensorProduct
    torch.manual_seed(1)
    a = L1TensorProduct("16x0e+16x1o", "24x0e+8x1o").to("cuda:0")
    b = SHTensorProduct("16x0e+16x1o", "24x0e+8x1o", 1).to("cuda:0")
    with torch.no_grad():
        b.weights_l0e.copy_(a.weights_l0e)
        b.weights_l1o.copy_(a.weights_l1o)
        assert torch.allclose(a.norm_l0e, b.norm_l0e) and torch.allclose(a.norm_l1o, b.norm_l1o)
        x, y = torch.randn(500, 64, device="cuda:0"), torch.randn(500, 4, device="cuda:0")
        assert ((a(x, y) - b(x, y)).abs().max() / a(x, y).abs().max()).item() < 1e-5


@pytest.mark.gpu
def test_gpu_equivariance_l2():
    from scalable_e3_gnn_amd.tensor_product import SHTensorProduct
    torch.manual_seed(2)
    irreps = "4x0e+3x1o+2x2e+1x1e"
    mod = SHTensorProduct(irreps, "3x0e+2x1o+2x2e+1x0o+1x1e+1x2o", 2).double().to("cuda:0")
    rng = np.random.default_rng(3)
    R = CG.random_rotation(rng)
    B = 50
    x = rng.normal(size=(B, mod.in1_dim))
    rel = rng.normal(size=(B, 3))
    def rot_feat(v, irr):
        v = v.copy(); col = 0
        for l, p, mul in O1.parse_blocks(irr):
            w = 2 * l + 1
            if l > 0:
                D = CG.rotation_matrices(l, R)
                v[:, col:col + w * mul] = (v[:, col:col + w * mul].reshape(-1, mul, w) @ D.T).reshape(-1, w * mul)
            col += w * mul
        return v
    f = lambda xx, rr: mod(torch.tensor(xx, device="cuda:0"), torch.tensor(CG.sh_component(2, rr), device="cuda:0")).cpu().numpy()
    with torch.no_grad():
        o1 = f(x, rel)
        o2 = f(rot_feat(x, irreps), rel @ R.T)
    assert np.abs(o2 - rot_feat(o1, "3x0e+2x1o+2x2e+1x0o+1x1e+1x2o")).max() < 1e-11


@pytest.mark.gpu
@pytest.mark.parametrize("lmax", [1, 2])
def test_gpu_fused_gather_gate_equals_unfused(lmax):
    """e3_tp_forward_fused (row gather + concat + TP + gate in one MFMA kernel) vs the unfused chain."""
    from scalable_e3_gnn_amd import ops
    from scalable_e3_gnn_amd.segnn import SEGNNLayer
    torch.manual_seed(3)
    H, N, E = 32, 500, 4001
    layer = SEGNNLayer(H, lmax).to("cuda:0")
    D = H * (4 if lmax == 1 else 9)
    ny = (lmax + 1) ** 2
    g = torch.Generator(device="cuda:0").manual_seed(4)
    h = torch.randn(N, D, device="cuda:0", generator=g)
    dst = torch.sort(torch.randint(0, N, (E,), device="cuda:0", generator=g)).values.int()
    src = torch.randint(0, N, (E,), device="cuda:0", generator=g).int()
    d = torch.rand(E, device="cuda:0", generator=g)
    Y = torch.randn(E, ny, device="cuda:0", generator=g)
    with torch.no_grad():
        assert layer.msg1.fused_supported(True) and layer.msg2.fused_supported(True) and layer.upd1.fused_supported(True)
        cat = torch.cat([h[dst.long()], h[src.long()], d[:, None]], 1)
        ref1 = layer._gate(layer.msg1(cat, Y))
        got1 = layer.msg1.forward_fused([(h, dst), (h, src), (d, None)], Y, gate=True)
        assert ((got1 - ref1).abs().max() / ref1.abs().max()).item() < 1e-5
        ref2 = layer._gate(layer.msg2(ref1, Y))
        got2 = layer.msg2.forward_fused([(ref1, None)], Y, gate=True)
        assert ((got2 - ref2).abs().max() / ref2.abs().max()).item() < 1e-5
        # no gate, single segment == plain forward (for l_max=1 `plain` is the exact-fp32 L1TP kernel, `raw` the
        # bf16-split MFMA kernel: both are within 1e-5 of the fp64 oracle, so within 1e-5 of each other)
        raw = layer.msg2.forward_fused([(ref1, None)], Y, gate=False)
        plain = layer.msg2(ref1, Y)
        assert ((raw - plain).abs().max() / plain.abs().max()).item() < 1e-5


@pytest.mark.gpu
def test_gpu_mfma_tp_vs_generic_kernel_large():
    """MFMA kernel (default) and generic FMA kernel (``exact``) against the fp64 module on a batch with a ragged tail."""
    from scalable_e3_gnn_amd.tensor_product import SHTensorProduct
    torch.manual_seed(7)
    a = SHTensorProduct("32x0e+32x1o+32x2e", "32x0e+64x0e+32x1o+32x2e", 2).to("cuda:0")
    b = SHTensorProduct("32x0e+32x1o+32x2e", "32x0e+64x0e+32x1o+32x2e", 2).double().to("cuda:0")
    b.load_state_dict({k: v.double() for k, v in a.state_dict().items()})
    x = torch.randn(10007, 288, device="cuda:0")
    y = torch.randn(10007, 9, device="cuda:0")
    with torch.no_grad():
        o32, o64 = a(x, y), b(x.double(), y.double())
        a.exact = True
        oex = a(x, y)
    e_mfma = ((o32.double() - o64).abs().max() / o64.abs().max()).item()
    e_exact = ((oex.double() - o64).abs().max() / o64.abs().max()).item()
    print(f"\nmessage TP 288 -> 352, B=10007 vs fp64: fp16-split MFMA {e_mfma:.2e}, exact fp32 FMA {e_exact:.2e}")
    assert e_mfma < 2e-6 and e_exact < 2e-6


@pytest.mark.gpu
def test_gpu_fused_scattered_out_blocks_vs_oracle():
    """A 32-channel output tile whose channels belong to two irreps blocks separated by another class (scattered
    columns): the MFMA kernel's per-channel column lookup path, checked against the fp64 oracle."""
    from scalable_e3_gnn_amd.tensor_product import SHTensorProduct
    torch.manual_seed(11)
    in1, out = "8x0e+8x1o", "8x0e+8x1o+8x0e"
    mod = SHTensorProduct(in1, out, 1).to("cuda:0")
    assert mod.fused_supported(False)
    B = 777
    g = torch.Generator().manual_seed(2)
    x = torch.randn(B, mod.in1_dim, generator=g, dtype=torch.float64)
    y = torch.randn(B, mod.in2_dim, generator=g, dtype=torch.float64)
    W = {c: getattr(mod, "weights_" + c).detach().double().cpu().numpy() for c in T.CLASSES if hasattr(mod, "weights_" + c)}
    N = {c: getattr(mod, "norm_" + c).double().cpu().numpy() for c in T.CLASSES}
    want = T.forward(in1, out, 1, x.numpy(), y.numpy(), W, N)
    with torch.no_grad():
        got = mod.forward_fused([(x.float().to("cuda:0"), None)], y.float().to("cuda:0"), gate=False)
    err = np.abs(got.double().cpu().numpy() - want).max() / np.abs(want).max()
    assert err < 1e-5, err


@pytest.mark.gpu
@pytest.mark.parametrize("io", ["float32", "bfloat16"])
def test_gpu_fused_unaligned_output_rows(io):
    """C ABI called with an output whose row stride is not a multiple of 4 elements (and a ragged last tile): the
    kernel must leave its 16-byte store path and still produce exactly the rows of the aligned call, touching
    nothing between the rows."""
    import ctypes
    from scalable_e3_gnn_amd import _lib
    from scalable_e3_gnn_amd.segnn import SEGNNLayer
    from scalable_e3_gnn_amd.tensor_product import TPSegment
    torch.manual_seed(5)
    dt = getattr(torch, io)
    layer = SEGNNLayer(32, 2).to("cuda:0").to(dt)
    tp = layer.msg2
    B = 32 * 5 + 7
    g = torch.Generator(device="cuda:0").manual_seed(6)
    m = torch.randn(B, 288, device="cuda:0", generator=g).to(dt)
    Y = torch.randn(B, 9, device="cuda:0", generator=g)
    from scalable_e3_gnn_amd import ops
    sc = ops.pow2_scale([m]) if dt == torch.float32 else None
    with torch.no_grad():
        ref = tp.forward_fused([(m, None)], Y, gate=True, in_scale=sc)
    lib = _lib.load()
    plan = tp._plan
    ws, ns = tp._tensors()
    packed = plan.packed(ws, ns, dt, m.device)
    ld = 288 + 3
    buf = torch.full((B, ld), 7.0, device="cuda:0", dtype=dt)
    segs = (TPSegment * 1)()
    segs[0].base, segs[0].ld, segs[0].ncols = m.data_ptr(), m.stride(0), 288
    stream = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.e3_tp_forward_fused(plan.handle(m.device), ctypes.byref(segs), 1, Y.data_ptr(), Y.stride(0),
                                       packed.data_ptr(), buf.data_ptr(), ld, B, _lib.dtype_code(dt), 1,
                                       sc.data_ptr() if sc is not None else None, stream), "e3_tp_forward_fused")
    torch.cuda.synchronize()
    assert torch.equal(buf[:, :288], ref)
    assert bool((buf[:, 288:] == 7.0).all())


@pytest.mark.gpu
@pytest.mark.parametrize("E", [1, 31, 33, 32 * 3, 32 * 4 + 5])
@pytest.mark.parametrize("io", ["float32", "bfloat16"])
def test_gpu_fused_message_products_tail_sizes(E, io):
    """Batch sizes around the 32-row tile (single partial tile, exact multiples, ragged tail) through the fused l_max=2
    message products (two-wave kernel: both waves must walk the same barriers on a partial tile) vs the unfused chain
    in fp64 on the same (storage-rounded) inputs."""
    from scalable_e3_gnn_amd.segnn import SEGNNLayer
    torch.manual_seed(100 + E)
    dt = getattr(torch, io)
    H, N = 32, 50
    layer = SEGNNLayer(H, 2).to("cuda:0")
    ref_layer = SEGNNLayer(H, 2).double().to("cuda:0")
    if io == "bfloat16":
        layer = layer.to(dt)
    ref_layer.load_state_dict({k: v.double() for k, v in layer.state_dict().items()})
    g = torch.Generator(device="cuda:0").manual_seed(5)
    h = torch.randn(N, 288, device="cuda:0", generator=g).to(dt)
    dst = torch.sort(torch.randint(0, N, (E,), device="cuda:0", generator=g)).values.int()
    src = torch.randint(0, N, (E,), device="cuda:0", generator=g).int()
    d = torch.rand(E, device="cuda:0", generator=g).to(dt)
    Y = torch.randn(E, 9, device="cuda:0", generator=g)
    with torch.no_grad():
        got1 = layer.msg1.forward_fused([(h, dst), (h, src), (d, None)], Y, gate=True)
        got2 = layer.msg2.forward_fused([(got1, None)], Y, gate=True)
        cat = torch.cat([h[dst.long()], h[src.long()], d[:, None]], 1).double()
        def gate64(t):  # [32 scalars | 32 gates(1o) | 32 gates(2e) | 32x1o | 32x2e] -> [silu | gated 1o | gated 2e]
            s, g1, g2, v1, v2 = t[:, :32], t[:, 32:64], t[:, 64:96], t[:, 96:192], t[:, 192:352]
            return torch.cat([torch.nn.functional.silu(s),
                              (torch.sigmoid(g1)[:, :, None] * v1.reshape(-1, 32, 3)).reshape(-1, 96),
                              (torch.sigmoid(g2)[:, :, None] * v2.reshape(-1, 32, 5)).reshape(-1, 160)], 1)
        ref1 = gate64(ref_layer.msg1(cat, Y.double()))
        ref2 = gate64(ref_layer.msg2(got1.double(), Y.double()))
    tol = 1e-5 if io == "float32" else 2e-2
    assert got1.shape == (E, 288) and got2.shape == (E, 288)
    assert ((got1.double() - ref1).abs().max() / ref1.abs().max()).item() < tol
    assert ((got2.double() - ref2).abs().max() / ref2.abs().max()).item() < tol
