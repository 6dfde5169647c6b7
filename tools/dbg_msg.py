import sys, numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import models  # noqa
from scalable_e3_gnn_amd import ops
from scalable_e3_gnn_amd.radius_graph import radius_graph
from scalable_e3_gnn_amd.segnn import SEGNNLayer
DEV = "cuda:0"
torch.manual_seed(5)
N = 900
pos = torch.rand(N, 3, generator=torch.Generator().manual_seed(4))
r = float((3 * 12.0 / (4 * np.pi * N)) ** (1 / 3))
g = radius_graph(pos.to(DEV), r, [0, 0, 0], [1, 1, 1])
layer = SEGNNLayer(32, 2).to(DEV)
Y, d, A = ops.edge_geometry(g, lmax=2)
for s in (1.0, 1e2, 1e3):
    h = torch.randn(N, 288, device=DEV) * s
    with torch.no_grad():
        for tp in (layer.msg1, layer.msg2): tp.exact = True
        m0 = ops.gather_concat(h, g, d)
        t1 = layer.msg1(m0, Y); m1 = layer._gate(t1)
        t2 = layer.msg2(m1, Y); m2 = layer._gate(t2)
        want = ops.segment_sum(m2, g)
        for tp in (layer.msg1, layer.msg2): tp.exact = False
        r1 = layer.msg1.forward_fused([(h, g.dst), (h, g.src), (d, None)], Y, gate=True)
        r2 = layer.msg2.forward_fused([(r1, None)], Y, gate=True)
        wr = ops.segment_sum(r2, g)
        got = layer._msg.forward(h, g, layer.msg1, layer.msg2)
    sc = want.abs().max()
    print(f"s={s}: fused-vs-exact {float((got-want).abs().max()/sc):.2e}  r16-vs-exact {float((wr-want).abs().max()/sc):.2e}  fused-vs-r16 {float((got-wr).abs().max()/sc):.2e}")
    print("   r16 m1 vs exact m1", float((r1-m1).abs().max()/m1.abs().max()), " raw t1 scale", float(t1.abs().max()), "m1 scale", float(m1.abs().max()))
    for name, lo, hi in (("0e", 0, 32), ("1o", 32, 128), ("2e", 128, 288)):
        print("   block", name, float((got[:, lo:hi]-want[:, lo:hi]).abs().max()/sc))
