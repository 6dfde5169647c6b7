"""Correctness probes AT THE BENCHMARKED SIZE (1 M particles, E ~ 23.5 M edges, l_max = 2, H = 32): sampled rows /
nodes of the dominant launches against the exact-fp32 generic kernels on the same inputs.  At this size the premix
table is 6.8 GB and an [E, 288] fp32 message buffer 27 GB, i.e. byte offsets pass 2^32 and 2^34 -- the index paths
that small-case parity tests never reach."""
import math

import numpy as np
import pytest
import torch

from scalable_e3_gnn_amd import ops
from scalable_e3_gnn_amd.radius_graph import radius_graph
from scalable_e3_gnn_amd.segnn import SEGNNLayer

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
N, H = 1_000_000, 32


@pytest.fixture(scope="module")
def big():
    torch.manual_seed(0)
    pos = torch.rand(N, 3, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1234))
    r = float((3.0 * 24.0 / (4.0 * math.pi * N)) ** (1.0 / 3.0))
    g = radius_graph(pos, r, [0, 0, 0], [1, 1, 1])
    layer = SEGNNLayer(H, 2).to(DEV)
    h = torch.randn(N, 288, device=DEV)
    Y, d, A = ops.edge_geometry(g, lmax=2)
    yield g, layer, h, Y, d
    del g, layer, h, Y, d
    torch.cuda.empty_cache()


def _exact(layer):
    for tp in (layer.msg1, layer.msg2):
        tp.exact = True


def _mfma(layer):
    for tp in (layer.msg1, layer.msg2):
        tp.exact = False


def test_fused_message_sampled_nodes_at_bench_size(big):
    g, layer, h, Y, d = big
    E = g.num_edges
    assert E > 20_000_000
    with torch.no_grad():
        a = layer._msg.forward(h, g, layer.msg1, layer.msg2)
    # first nodes (first tiles), last nodes (ragged last tile), nodes whose premix rows start beyond 2^32 bytes
    # (node > 633 k at 6784 B per node), and a random sample
    rp = g.rowptr.cpu().numpy().astype(np.int64)
    rng = np.random.default_rng(0)
    nodes = np.unique(np.concatenate([np.arange(0, 40), np.arange(N - 40, N), np.arange(633_200, 633_240),
                                      np.arange(950_000, 950_020), rng.integers(0, N, 400)]))
    eidx = np.concatenate([np.arange(rp[i], rp[i + 1]) for i in nodes])
    seg = np.concatenate([np.full(rp[i + 1] - rp[i], k) for k, i in enumerate(nodes)])
    ei = torch.as_tensor(eidx, device=DEV)
    dst = torch.as_tensor(np.repeat(nodes, rp[nodes + 1] - rp[nodes]), device=DEV)
    src = g.src[ei].long()
    _exact(layer)
    with torch.no_grad():
        m = torch.cat([h[dst], h[src], d[ei].unsqueeze(1)], 1)
        m = layer._gate(layer.msg1(m, Y[ei]))
        m = layer._gate(layer.msg2(m, Y[ei]))
        want = torch.zeros(len(nodes), 288, device=DEV, dtype=torch.float64)
        want.index_add_(0, torch.as_tensor(seg, device=DEV), m.double())
    _mfma(layer)
    got = a[torch.as_tensor(nodes, device=DEV)].double()
    err = float((got - want).abs().max() / want.abs().max())
    assert err < 2e-6, err
    assert torch.isfinite(a).all()


def test_fused_message_bf16_sampled_nodes_at_bench_size(big):
    """The bf16-storage leg of the bench (configs[2]) at its own size: the same sampled nodes as the fp32 check, against the
    exact fp32 chain evaluated on the bf16-rounded features and weights.  The messages between the two products are rounded
    to bf16 in the kernel (8 significant bits), so the bound is a bf16 one; the measured figure is printed."""
    g, layer, h, Y, d = big
    layer16 = SEGNNLayer(H, 2).to(DEV)
    layer16.load_state_dict(layer.state_dict())
    layer16 = layer16.bfloat16()
    h16 = h.bfloat16()
    with torch.no_grad():
        a = layer16._msg.forward(h16, g, layer16.msg1, layer16.msg2)
    assert torch.isfinite(a).all()
    rp = g.rowptr.cpu().numpy().astype(np.int64)
    rng = np.random.default_rng(0)
    nodes = np.unique(np.concatenate([np.arange(0, 40), np.arange(N - 40, N), np.arange(633_200, 633_240),
                                      np.arange(950_000, 950_020), rng.integers(0, N, 400)]))
    eidx = np.concatenate([np.arange(rp[i], rp[i + 1]) for i in nodes])
    seg = np.concatenate([np.full(rp[i + 1] - rp[i], k) for k, i in enumerate(nodes)])
    ei = torch.as_tensor(eidx, device=DEV)
    dst = torch.as_tensor(np.repeat(nodes, rp[nodes + 1] - rp[nodes]), device=DEV)
    src = g.src[ei].long()
    ref = SEGNNLayer(H, 2).to(DEV)   # fp32 layer holding the bf16-rounded parameters
    ref.load_state_dict({k: v.float() for k, v in layer16.state_dict().items()})
    _exact(ref)
    hr = h16.float()
    with torch.no_grad():
        m = torch.cat([hr[dst], hr[src], d[ei].unsqueeze(1)], 1)
        m = ref._gate(ref.msg1(m, Y[ei]))
        m = ref._gate(ref.msg2(m, Y[ei]))
        want = torch.zeros(len(nodes), 288, device=DEV, dtype=torch.float64)
        want.index_add_(0, torch.as_tensor(seg, device=DEV), m.double())
    got = a[torch.as_tensor(nodes, device=DEV)].double()
    err = float((got - want).abs().max() / want.abs().max())
    print(f"\nbf16-storage fused message at 1 M particles, sampled nodes vs exact fp32 chain on bf16-rounded operands: {err:.2e}")
    assert err < 3e-3, err   # measured 8.0e-4 (round 3)
    del layer16, ref, a
    torch.cuda.empty_cache()


def test_r16_message_tp1_sampled_rows_at_bench_size(big):
    """The per-TP fused kernel (gather + concat + TP + gate, used for bf16 storage and with fuse_message = False) writes
    [E, 288] fp32 = 27 GB: rows of the first tile, of the last (ragged) tile, beyond 2^32 and beyond 2^34 bytes."""
    g, layer, h, Y, d = big
    E = g.num_edges
    with torch.no_grad():
        m = layer.msg1.forward_fused([(h, g.dst), (h, g.src), (d, None)], Y, gate=True)
    assert m.shape == (E, 288)
    rows = np.unique(np.concatenate([np.arange(0, 32), np.arange(E - 37, E), np.arange(3_728_300, 3_728_340),
                                     np.arange(14_913_100, 14_913_140), np.arange(20_000_000, 20_000_020)]))
    assert rows[-1] * 288 * 4 > 2 ** 34 > 3_728_300 * 288 * 4 > 2 ** 32
    ri = torch.as_tensor(rows, device=DEV)
    _exact(layer)
    with torch.no_grad():
        x = torch.cat([h[g.dst[ri].long()], h[g.src[ri].long()], d[ri].unsqueeze(1)], 1)
        want = layer._gate(layer.msg1(x, Y[ri])).double()
    _mfma(layer)
    err = float((m[ri].double() - want).abs().max() / want.abs().max())
    assert err < 2e-6, err
    del m
    torch.cuda.empty_cache()
