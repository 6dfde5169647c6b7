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
if os.environ.get("DTYPE") == "bf16":
    layer = layer.bfloat16(); h = h.bfloat16(); d = d.bfloat16()
dst = g.dst
src_ix = g.src
if os.environ.get('ZERO_IDX'):  # every gather hits the same few rows: isolates instruction cost from memory cost
    dst = torch.zeros_like(dst); src_ix = torch.zeros_like(src_ix)
def t(fn, n=20):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): out = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, out
with torch.no_grad():
    m1t, m = t(lambda: layer.msg1.forward_fused([(h, dst), (h, src_ix), (d, None)], Y, gate=True))
    m2t, m2 = t(lambda: layer.msg2.forward_fused([(m, None)], Y, gate=True))
E = g.num_edges
print(f"N={N} E={E} lmax={lmax} dbg={os.environ.get('E3_TP_DBG','0')} nbuf={os.environ.get('E3_TP_NBUF','-')} exact={os.environ.get('E3_TP_EXACT','0')} dtype={os.environ.get('DTYPE','f32')}: "
      f"msg1 {m1t:.2f} ms ({m1t*1e6/E*32/1e3:.1f} us/tile-wave... {E/m1t/1e3:.0f} Medges/s)  msg2 {m2t:.2f} ms ({E/m2t/1e3:.0f} Medges/s)")

if int(os.environ.get("E3_TP_DBG", "0")) & 8:
    import ctypes
    from scalable_e3_gnn_amd import _lib
    lib = _lib.load()
    for name, tp, call in (("msg1", layer.msg1, lambda: layer.msg1.forward_fused([(h, dst), (h, src_ix), (d, None)], Y, gate=True)),
                           ("msg2", layer.msg2, lambda: layer.msg2.forward_fused([(m, None)], Y, gate=True))):
        plan = tp._plan if hasattr(tp, "_plan") and hasattr(tp._plan, "handle") else tp._fused_plan()
        buf = (ctypes.c_ulonglong * 8)()
        lib.e3_tp_debug_phase_cycles(plan.handle, buf)  # clear
        with torch.no_grad():
            call()
        lib.e3_tp_debug_phase_cycles(plan.handle, buf)
        tot = sum(buf)
        print(name, "phase shares: prologue %.1f%% wait %.1f%% stage-issue %.1f%% runs %.1f%% epi-transpose %.1f%% epi-store %.1f%% [stage-setup|partner-wait %.1f%% w-preload %.1f%%]" % tuple(100.0 * b / tot for b in list(buf)[:8]),
              " cycles/tile-wave %.0f" % (tot / (E / 32)))
