#!/usr/bin/env python3
"""Fused message-TP micro-benchmark (development tool): times msg1 / msg2 of one SEGNN layer on a synthetic graph."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import models
from scalable_e3_gnn_amd import ops
from scalable_e3_gnn_amd.radius_graph import radius_graph
from scalable_e3_gnn_amd.segnn import SEGNNLayer
N = int(os.environ.get("N", 300000)); lmax = int(os.environ.get("LMAX", 2)); H = 32
dev = "cuda:0"
torch.manual_seed(0)
pos = torch.rand(N, 3, device=dev)
r = float((3 * 24 / (4 * np.pi * N)) ** (1 / 3))
g = radius_graph(pos, r, [0, 0, 0], [1, 1, 1])
layer = SEGNNLayer(H, lmax).to(dev)
D = H * (4 if lmax == 1 else 9)
h = torch.randn(N, D, device=dev)
Y, d, A = ops.edge_geometry(g, lmax=lmax)
dst = g.dst
def t(fn, n=20):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): out = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, out
with torch.no_grad():
    m1t, m = t(lambda: layer.msg1.forward_fused([(h, dst), (h, g.src), (d, None)], Y, gate=True))
    m2t, m2 = t(lambda: layer.msg2.forward_fused([(m, None)], Y, gate=True))
E = g.num_edges
print(f"N={N} E={E} lmax={lmax} dbg={os.environ.get('E3_TP_DBG','0')} nbuf={os.environ.get('E3_TP_NBUF','-')} exact={os.environ.get('E3_TP_EXACT','0')}: "
      f"msg1 {m1t:.2f} ms ({m1t*1e6/E*32/1e3:.1f} us/tile-wave... {E/m1t/1e3:.0f} Medges/s)  msg2 {m2t:.2f} ms ({E/m2t/1e3:.0f} Medges/s)")
