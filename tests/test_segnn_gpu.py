"""SEGNN forward (graph build -> edge geometry -> L layers -> readout) on the GPU vs the numpy fp64
oracle.  All stages except the tensor product are builder-defined ("parity unpinned" w.r.t. upstream).
Tolerance: 1e-5 relative to the output scale per stage input->output (north_star), 1e-5 end to end
after L layers of fp32 accumulation."""
import numpy as np
import pytest
import torch

from oracle import graph_oracle as G
from oracle import segnn_oracle as S
from scalable_e3_gnn_amd import ops
from scalable_e3_gnn_amd.radius_graph import radius_graph
from scalable_e3_gnn_amd.segnn import SEGNN

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel(a, b):
    a = a.detach().double().cpu().numpy() if isinstance(a, torch.Tensor) else a
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def setup(N, H, L, seed=0, k=12.0):
    torch.manual_seed(seed)
    pos = torch.rand(N, 3, generator=torch.Generator().manual_seed(seed))
    r = float((3 * k / (4 * np.pi * N)) ** (1 / 3))
    model = SEGNN("1x0e+1x1o", H, "1x1o", L).to(DEV)
    g = radius_graph(pos.to(DEV), r, [0, 0, 0], [1, 1, 1])
    x = torch.randn(N, 4, generator=torch.Generator().manual_seed(seed + 1))
    xs = x[g.perm.cpu().long()]
    return model, g, xs, pos, r


def test_edge_ops_vs_oracle():
    model, g, xs, pos, r = setup(3000, 8, 1)
    perm, rp, src = G.graph(pos.numpy(), [0, 0, 0], [1, 1, 1], r)
    assert np.array_equal(g.src.cpu().numpy(), src)
    Y, d, A = ops.edge_geometry(g)
    Yo, do, Ao, dst = S.edge_geometry(pos.numpy()[perm], rp, src)
    assert rel(Y, Yo) < 1e-5 and rel(d, do) < 1e-6 and rel(A, Ao) < 1e-5
    h = torch.randn(3000, 32, device=DEV)
    m = ops.gather_concat(h, g, d)
    hn = h.cpu().numpy()
    assert np.array_equal(m.cpu().numpy(), np.concatenate([hn[dst], hn[src], d.cpu().numpy()[:, None]], 1))
    t = torch.randn(500, 8 + 8 + 24, device=DEV)
    assert rel(ops.gate(t, 8, 8), S.gate(t.double().cpu().numpy(), 8, 8)) < 1e-6
    msg = torch.randn(g.num_edges, 32, device=DEV)
    agg = ops.segment_sum(msg, g)
    want = np.zeros((3000, 32))
    np.add.at(want, dst, msg.double().cpu().numpy())
    assert rel(agg, want) < 1e-6
    assert torch.equal(agg, ops.segment_sum(msg, g))  # bitwise reproducible


@pytest.mark.parametrize("N,H,L", [(1000, 8, 2), (2000, 32, 4)])
def test_segnn_forward_vs_oracle(N, H, L):
    model, g, xs, pos, r = setup(N, H, L, seed=2)
    out = model(xs.to(DEV), g)
    params = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    perm = g.perm.cpu().numpy()
    want, trace = S.forward(params, H, L, "1x0e+1x1o", "1x1o", xs.double().numpy(), pos.numpy()[perm],
                            g.rowptr.cpu().numpy(), g.src.cpu().numpy(), return_all=True)
    assert out.shape == want.shape
    assert rel(out, want) < 1e-5, rel(out, want)


def test_segnn_equivariance():
    """Rotate + translate the cloud and the input vectors: scalars invariant, vectors co-rotate."""
    N, H, L = 1500, 16, 2
    model, g, xs, pos, r = setup(N, H, L, seed=4)
    out = model(xs.to(DEV), g)
    q, _ = torch.linalg.qr(torch.randn(3, 3, dtype=torch.float64))
    R = (q * torch.sign(torch.linalg.det(q))).float()
    pos2 = (pos - 0.5) @ R.T + 0.5
    g2 = radius_graph(pos2.to(DEV), r, [-0.5, -0.5, -0.5], [1.5, 1.5, 1.5])
    x = torch.empty(N, 4)
    x[g.perm.cpu().long()] = xs
    x2 = x.clone()
    x2[:, 1:] = x[:, 1:] @ R.T
    out2 = model(x2[g2.perm.cpu().long()].to(DEV), g2)
    o = torch.empty(N, 3)
    o[g.perm.cpu().long()] = out.cpu()
    o2 = torch.empty(N, 3)
    o2[g2.perm.cpu().long()] = out2.cpu()
    # a handful of pairs sit within fp32 rounding of the cutoff and may flip; compare the bulk
    err = ((o2 - o @ R.T).abs().max(1).values / o.abs().max())
    assert (err < 1e-3).float().mean() > 0.99 and err.median() < 1e-5


def test_segnn_lmax2_forward_vs_oracle():
    """l_max = 2 (the BASELINE headline configuration's operator set) at a size the oracle finishes in seconds."""
    N, H, L = 800, 8, 2
    torch.manual_seed(5)
    pos = torch.rand(N, 3, generator=torch.Generator().manual_seed(5))
    r = float((3 * 12.0 / (4 * np.pi * N)) ** (1 / 3))
    model = SEGNN("1x0e+1x1o", H, "1x1o", L, lmax=2).to(DEV)
    g = radius_graph(pos.to(DEV), r, [0, 0, 0], [1, 1, 1])
    x = torch.randn(N, 4, generator=torch.Generator().manual_seed(6))
    xs = x[g.perm.cpu().long()]
    with torch.no_grad():
        out = model(xs.to(DEV), g)
    Y, d, A = ops.edge_geometry(g, lmax=2)
    perm = g.perm.cpu().numpy()
    Yo, do, Ao, _ = S.edge_geometry_l2(pos.numpy()[perm], g.rowptr.cpu().numpy(), g.src.cpu().numpy())
    assert rel(Y, Yo) < 1e-5 and rel(A, Ao) < 1e-5
    params = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    want = S.forward_l2(params, H, L, "1x0e+1x1o", "1x1o", xs.double().numpy(), pos.numpy()[perm],
                        g.rowptr.cpu().numpy(), g.src.cpu().numpy())
    assert rel(out, want) < 1e-5, rel(out, want)


def test_segnn_lmax2_fused_H32_vs_oracle():
    """H = 32 takes the fused MFMA path (gather + TP + gate in one kernel); small cloud, 1 layer."""
    N, H, L = 300, 32, 1
    torch.manual_seed(8)
    pos = torch.rand(N, 3, generator=torch.Generator().manual_seed(8))
    r = float((3 * 10.0 / (4 * np.pi * N)) ** (1 / 3))
    model = SEGNN("1x0e+1x1o", H, "1x1o", L, lmax=2).to(DEV)
    g = radius_graph(pos.to(DEV), r, [0, 0, 0], [1, 1, 1])
    xs = torch.randn(N, 4, generator=torch.Generator().manual_seed(9))[g.perm.cpu().long()]
    with torch.no_grad():
        assert model.layers[0].fused_available()
        out = model(xs.to(DEV), g)
        model.layers[0].fused = False
        out_unfused = model(xs.to(DEV), g)
    assert rel(out, out_unfused.double().cpu().numpy()) < 1e-5
    params = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    perm = g.perm.cpu().numpy()
    want = S.forward_l2(params, H, L, "1x0e+1x1o", "1x1o", xs.double().numpy(), pos.numpy()[perm],
                        g.rowptr.cpu().numpy(), g.src.cpu().numpy())
    assert rel(out, want) < 1e-5, rel(out, want)


@pytest.mark.gpu
def test_segnn_lmax1_fused_H32_vs_oracle():
    """l_max = 1, H = 32 under no_grad: the fused path on the reference operator's plans (one-wave MFMA kernel, no
    fused segment-sum instantiation -> the scatter request must fall back to the two kernels)."""
    N, H, L = 400, 32, 2
    torch.manual_seed(12)
    pos = torch.rand(N, 3, generator=torch.Generator().manual_seed(12))
    r = float((3 * 10.0 / (4 * np.pi * N)) ** (1 / 3))
    model = SEGNN("1x0e+1x1o", H, "1x1o", L, lmax=1).to(DEV)
    g = radius_graph(pos.to(DEV), r, [0, 0, 0], [1, 1, 1])
    xs = torch.randn(N, 4, generator=torch.Generator().manual_seed(13))[g.perm.cpu().long()]
    with torch.no_grad():
        assert model.layers[0].fused_available()
        out = model(xs.to(DEV), g)
    params = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    perm = g.perm.cpu().numpy()
    want = S.forward(params, H, L, "1x0e+1x1o", "1x1o", xs.double().numpy(), pos.numpy()[perm],
                     g.rowptr.cpu().numpy(), g.src.cpu().numpy())
    assert rel(out, want) < 1e-5, rel(out, want)


@pytest.mark.gpu
@pytest.mark.parametrize("io", ["float32", "bfloat16"])
def test_fused_segment_sum_matches_two_kernels(io):
    """e3_tp_forward_fused_scatter (message TP #2 with the segment-sum in its epilogue, fp32 atomics) vs
    e3_tp_forward_fused + e3_segment_sum: equal to fp32 rounding of the sums (the order of the atomics is not fixed);
    bf16 storage: the two-kernel path rounds every message to bf16 before summing, the fused one does not, so they
    agree to bf16 resolution of the messages."""
    import torch
    from scalable_e3_gnn_amd import ops
    from scalable_e3_gnn_amd.radius_graph import radius_graph
    from scalable_e3_gnn_amd.segnn import SEGNNLayer
    torch.manual_seed(21)
    dev = "cuda:0"
    dt = getattr(torch, io)
    N = 3000
    pos = torch.rand(N, 3, device=dev)
    g = radius_graph(pos, 0.12, [0, 0, 0], [1, 1, 1])
    E = g.num_edges
    layer = SEGNNLayer(32, 2).to(dev).to(dt)
    m = torch.randn(E, 288, device=dev).to(dt)
    Y = torch.randn(E, 9, device=dev)
    with torch.no_grad():
        ref = ops.segment_sum(layer.msg2.forward_fused([(m, None)], Y, gate=True), g)
        got = layer.msg2.forward_fused([(m, None)], Y, gate=True, scatter=(g.dst, N))
    if io == "bfloat16" and got is None:
        pytest.skip("fused segment-sum is not instantiated for bf16 storage (slower than the two kernels)")
    assert got is not None, "fused scatter kernel missing for the l_max=2 message product"
    assert got.shape == ref.shape and got.dtype == ref.dtype
    err = (got.float() - ref.float()).abs().max().item() / ref.float().abs().max().item()
    assert err < (1e-5 if io == "float32" else 2e-2), err
    # nodes without edges stay exactly zero
    deg = (g.rowptr[1:] - g.rowptr[:-1])
    assert bool((got[deg == 0] == 0).all())
