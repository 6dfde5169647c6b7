"""bf16 storage path (BASELINE config 3: "bf16 on 1 MI355X"): bf16 features / weights / outputs, fp32 spherical
harmonics, one bf16 MFMA product with fp32 accumulation.  Checked against the fp64 oracle evaluated on the SAME
bf16-rounded inputs and weights.  Tolerances (relative to the output scale): one TP 1e-2 (the per-row feature and
the result are each rounded once to 8 significant bits), a 2-layer network 5e-2."""
import numpy as np
import pytest
import torch

from oracle import segnn_oracle as S
from oracle import tp_oracle as T

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel(a, b):
    a = a.detach().double().cpu().numpy() if isinstance(a, torch.Tensor) else a
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.mark.parametrize("in1,out", [
    ("32x0e+32x1o+32x2e", "32x0e+64x0e+32x1o+32x2e"),
    ("32x0e+32x1o+32x2e", "32x0e+32x1o+32x2e"),
    ("1x0e+1x1o", "32x0e+32x1o+32x2e"),
    ("32x0e+32x1o+32x2e", "1x1o"),
])
def test_bf16_tp_vs_oracle(in1, out):
    from scalable_e3_gnn_amd.tensor_product import SHTensorProduct
    torch.manual_seed(0)
    mod = SHTensorProduct(in1, out, 2).bfloat16().to(DEV)
    B = 1003
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, mod.in1_dim, generator=g).bfloat16()
    y = torch.randn(B, 9, generator=g)
    W = {c: getattr(mod, "weights_" + c).detach().float().double().cpu().numpy() for c in T.CLASSES if hasattr(mod, "weights_" + c)}
    N = {c: getattr(mod, "norm_" + c).float().double().cpu().numpy() for c in T.CLASSES}
    want = T.forward(in1, out, 2, x.float().double().numpy(), y.double().numpy(), W, N)
    with torch.no_grad():
        got = mod(x.to(DEV), y.to(DEV))
    assert got.dtype == torch.bfloat16 and got.shape == want.shape
    assert rel(got.float(), want) < 1e-2, rel(got.float(), want)


def test_bf16_fused_message_tp_vs_oracle():
    from scalable_e3_gnn_amd.segnn import SEGNNLayer
    torch.manual_seed(3)
    H, N, E = 32, 400, 3001
    layer = SEGNNLayer(H, 2).bfloat16().to(DEV)
    g = torch.Generator().manual_seed(4)
    h = torch.randn(N, 288, generator=g).bfloat16()
    dst = torch.sort(torch.randint(0, N, (E,), generator=g)).values.int()
    src = torch.randint(0, N, (E,), generator=g).int()
    d = torch.rand(E, generator=g).bfloat16()
    Y = torch.randn(E, 9, generator=g)
    with torch.no_grad():
        got = layer.msg1.forward_fused([(h.to(DEV), dst.to(DEV)), (h.to(DEV), src.to(DEV)), (d.to(DEV), None)], Y.to(DEV), gate=True)
    assert got.dtype == torch.bfloat16 and got.shape == (E, 288)
    tp = layer.msg1
    W = {c: getattr(tp, "weights_" + c).detach().float().double().cpu().numpy() for c in T.CLASSES if hasattr(tp, "weights_" + c)}
    Nn = {c: getattr(tp, "norm_" + c).float().double().cpu().numpy() for c in T.CLASSES}
    hd = h.float().double().numpy()
    cat = np.concatenate([hd[dst.long().numpy()], hd[src.long().numpy()], d.float().double().numpy()[:, None]], 1)
    hid, gated = "32x0e+32x1o+32x2e", "32x0e+64x0e+32x1o+32x2e"
    raw = T.forward(f"{hid}+{hid}+1x0e", gated, 2, cat, Y.double().numpy(), W, Nn)
    want = S.gate_blocks(raw, H, [(1, H), (2, H)])
    assert rel(got.float(), want) < 1e-2, rel(got.float(), want)


def test_bf16_segnn_forward_vs_oracle():
    from scalable_e3_gnn_amd.radius_graph import radius_graph
    from scalable_e3_gnn_amd.segnn import SEGNN
    N, H, L = 400, 32, 2
    torch.manual_seed(5)
    pos = torch.rand(N, 3, generator=torch.Generator().manual_seed(5))
    r = float((3 * 10.0 / (4 * np.pi * N)) ** (1 / 3))
    model = SEGNN("1x0e+1x1o", H, "1x1o", L, lmax=2).bfloat16().to(DEV)
    g = radius_graph(pos.to(DEV), r, [0, 0, 0], [1, 1, 1])
    xs = torch.randn(N, 4, generator=torch.Generator().manual_seed(6))[g.perm.cpu().long()].bfloat16()
    with torch.no_grad():
        out = model(xs.to(DEV), g)
    assert out.dtype == torch.bfloat16
    params = {k: v.detach().float().cpu().numpy() for k, v in model.state_dict().items()}
    perm = g.perm.cpu().numpy()
    want = S.forward_l2(params, H, L, "1x0e+1x1o", "1x1o", xs.float().double().numpy(), pos.numpy()[perm],
                        g.rowptr.cpu().numpy(), g.src.cpu().numpy())
    assert rel(out.float(), want) < 5e-2, rel(out.float(), want)


@pytest.mark.parametrize("lmax,H", [(2, 32), (1, 32), (2, 64)])
def test_bf16_fused_message_kernel_vs_oracle(lmax, H):
    """The one-launch message function in bf16 storage (e3_msg_forward, dtype E3_BF16) against the fp64 oracle chain on
    the same bf16-rounded features and weights: two products and two gates, messages rounded to bf16 in between."""
    from oracle import cg
    from scalable_e3_gnn_amd.radius_graph import radius_graph
    from scalable_e3_gnn_amd.segnn import SEGNNLayer
    torch.manual_seed(7 + H + lmax)
    N = 700
    pos = torch.rand(N, 3, generator=torch.Generator().manual_seed(8))
    r = float((3 * 12.0 / (4 * np.pi * N)) ** (1 / 3))
    g = radius_graph(pos.to(DEV), r, [0, 0, 0], [1, 1, 1])
    layer = SEGNNLayer(H, lmax).bfloat16().to(DEV)
    assert layer._msg.supports(torch.bfloat16)
    D = H * (lmax + 1) ** 2
    h = torch.randn(N, D, generator=torch.Generator().manual_seed(9)).bfloat16()
    with torch.no_grad():
        got = layer._msg.forward(h.to(DEV), g, layer.msg1, layer.msg2)
    assert got.dtype == torch.float32 and got.shape == (N, D)
    hid = f"{H}x0e+{H}x1o" + (f"+{H}x2e" if lmax == 2 else "")
    gated = f"{H}x0e+{lmax * H}x0e+{H}x1o" + (f"+{H}x2e" if lmax == 2 else "")
    blocks = [(l, H) for l in range(1, lmax + 1)]

    def WN(tp):
        W = {c: getattr(tp, "weights_" + c).detach().float().double().cpu().numpy() for c in T.CLASSES if hasattr(tp, "weights_" + c)}
        Nn = {c: (getattr(tp, "norm_" + c).float().double().cpu().numpy() if hasattr(tp, "norm_" + c) else np.zeros(0)) for c in T.CLASSES}
        return W, Nn

    src, dst = g.src.cpu().long().numpy(), g.dst.cpu().long().numpy()
    p64 = g.pos4[:, :3].double().cpu().numpy()
    rel_v = p64[src] - p64[dst]
    Y = cg.sh_component(lmax, rel_v)
    dd = np.sqrt((rel_v * rel_v).sum(1))
    hd = h.float().double().numpy()
    m = np.concatenate([hd[dst], hd[src], dd[:, None]], 1)
    m = S.gate_blocks(T.forward(f"{hid}+{hid}+1x0e", gated, lmax, m, Y, *WN(layer.msg1)), H, blocks)
    m = S.gate_blocks(T.forward(hid, gated, lmax, m, Y, *WN(layer.msg2)), H, blocks)
    want = np.zeros((N, D))
    np.add.at(want, dst, m)
    assert rel(got, want) < 2e-2, rel(got, want)
