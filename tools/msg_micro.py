#!/usr/bin/env python3
"""Fused message kernel micro-benchmark (development tool): e3_msg_forward of one SEGNN layer on a synthetic graph.
env: N (particles), LMAX, H, TPB (tiles per block list, comma separated), ZERO_IDX (every gather hits row 0), ITERS"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import models  # noqa
from scalable_e3_gnn_amd import _lib as _L
if os.environ.get("E3_EXP_LIB"):  # experiment build (tools/build_variant.sh), development only
    _L.LIB_PATH = os.path.join(os.path.dirname(_L.LIB_PATH), "exp", f"libe3gnn_{os.environ['E3_EXP_LIB']}.so")
from scalable_e3_gnn_amd import ops
from scalable_e3_gnn_amd.radius_graph import radius_graph
from scalable_e3_gnn_amd.segnn import SEGNNLayer
N = int(os.environ.get("N", 1000000)); lmax = int(os.environ.get("LMAX", 2)); H = int(os.environ.get("H", 32))
iters = int(os.environ.get("ITERS", 5))
dev = "cuda:0"
torch.manual_seed(0)
pos = torch.rand(N, 3, device=dev)
r = float((3 * 24 / (4 * np.pi * N)) ** (1 / 3))
g = radius_graph(pos, r, [0, 0, 0], [1, 1, 1])
layer = SEGNNLayer(H, lmax).to(dev)
D = H * (lmax + 1) ** 2
h = torch.randn(N, D, device=dev)
sc = ops.pow2_scale([h])
if os.environ.get("DT") == "bf16":   # bf16 storage (BASELINE.json configs[2]): no operand scales
    layer = layer.bfloat16(); h = h.bfloat16(); sc = None
edges = None
if os.environ.get("ZERO_IDX"):   # every h[src] gather hits row 0
    edges = (torch.zeros_like(g.src), g.dst)
if os.environ.get("ZERO_DST"):   # every pre-mix row is row 0 and nothing is ever flushed (one run): no U misses, no atomics
    edges = ((edges[0] if edges else g.src), torch.zeros_like(g.dst))
E = g.num_edges
for tpb in [int(v) for v in os.environ.get("TPB", "0").split(",")]:
    layer._msg.tiles_per_block = tpb
    with torch.no_grad():
        for _ in range(2):
            layer._msg.forward(h, g, layer.msg1, layer.msg2, sc, edges=edges)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            layer._msg.forward(h, g, layer.msg1, layer.msg2, sc, edges=edges)
        e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    if os.environ.get("STAMPS"):  # per-phase cycle counters of a -DE3_MSG_STAMP=1 build (development)
        import ctypes
        lib = ctypes.CDLL(_L.LIB_PATH)
        buf = (ctypes.c_ulonglong * 8)()
        lib.e3_msg_debug_stamps(buf, 1)
        layer._msg.forward(h, g, layer.msg1, layer.msg2, sc, edges=edges); torch.cuda.synchronize()
        lib.e3_msg_debug_stamps(buf, 0)
        names = ["gather issue + geometry", "gather wait", "product #1", "gate #1 + park", "product #2", "loop top (ids, reloads)",
                 "gate #2 + transposed writes", "run sums + flushes"]
        tot = sum(buf[:8])
        for nme, v in zip(names, buf[:8]):
            print(f"   stamp {nme:32s} {v/tot*100:5.1f} %   {v/((E+15)//16):9.0f} cycles per tile and wave")
    if os.environ.get("WS_STAMPS"):  # per-wave cycle counters of a -DE3_WS_STAMP=1 build of e3_msg_ws.hip (development)
        import ctypes
        lib = ctypes.CDLL(_L.LIB_PATH)
        buf = (ctypes.c_ulonglong * 96)()
        lib.e3_msg_ws_debug_stamps(buf, 1)
        layer._msg.forward(h, g, layer.msg1, layer.msg2, sc, edges=edges); torch.cuda.synchronize()
        lib.e3_msg_ws_debug_stamps(buf, 0)
        nwg = 256
        steps = (E / 15.7) / nwg   # ~tiles per workgroup
        print("   wave: cycles per step  [phase A work | barrier A | phase B work | barrier B + loop top]")
        for w in range(8):
            v = [buf[4 * w + i] / nwg / steps for i in range(4)]
            print(f"   wave {w}: {v[0]:8.0f} {v[1]:8.0f} {v[2]:8.0f} {v[3]:8.0f}   sum {sum(v):8.0f}")
        print("   tensor product marks, cycles per tile: [first requests | degree 2 | degree 1 (+fold 2) | degree 0 (+fold 1) | fold 0]  (rest of the phase = gate + stores)")
        for w in range(8):
            v = [buf[32 + 8 * w + i] / nwg / steps for i in range(8)]
            print(f"   wave {w}: " + " ".join(f"{x:7.0f}" for x in v[:5]) + f"   sum {sum(v[:5]):7.0f}   | before product {v[5]:6.0f}"
                  f" | X after product / all of X {v[6]:6.0f} | Y after product / all of Y {v[7]:6.0f}")
    fl = layer._msg.flops_per_edge() * E
    print(f"[{os.environ.get('E3_EXP_LIB','default')}] N={N} E={E} lmax={lmax} H={H} tpb={tpb} zero_idx={bool(edges)}: {ms:.2f} ms/launch pair  {E/ms/1e3:.0f} Medges/s  "
          f"{fl/ms/1e9:.1f} TFLOP/s algorithmic ({3*fl/ms/1e9/2500*100:.1f} % of bf16/f16 MFMA peak executed x3)")
