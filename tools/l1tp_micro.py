#!/usr/bin/env python3
"""L1TP operator micro-benchmark (rows/s, algorithmic GB/s) — development tool, not the judged bench."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import models  # noqa
from models.segnn.l1_tensor_prod import L1TensorProduct
from scalable_e3_gnn_amd import Irreps

ap = argparse.ArgumentParser()
ap.add_argument("--H", type=int, nargs="+", default=[32])
ap.add_argument("--B", type=int, default=1 << 21)
ap.add_argument("--kernels", type=int, nargs="+", default=[1, 2])
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--irreps", type=str, default=None)
ap.add_argument("--out", type=str, default=None)
a = ap.parse_args()
dev = "cuda:0"
for H in a.H:
    ir = a.irreps or f"{H}x0e+{H}x1o"
    torch.manual_seed(0)
    mod = L1TensorProduct(Irreps(ir), Irreps(a.out) if a.out else None).to(dev)
    x = torch.randn(a.B, mod.in1_dim, device=dev)
    y = torch.randn(a.B, 4, device=dev)
    Dout = mod.iro.dim
    ref = None
    for k in a.kernels:
        mod.kernel = k
        with torch.no_grad():
            try:
                o = mod(x, y)
            except RuntimeError as e:
                print(f"{ir} kernel={k}: {e}")
                continue
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                o = mod(x, y)
            e1.record()
            torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        gb = 4 * (mod.in1_dim + 4 + Dout) * a.B / 1e9
        err = 0.0 if ref is None else ((o - ref).abs().max() / ref.abs().max()).item()
        if ref is None:
            ref = o.clone()
        print(f"{ir} -> {a.out or ir}  B={a.B} kernel={k}: {ms:.3f} ms  {a.B / ms / 1e6:.1f} Grows/s*1e-3  "
              f"{gb / ms * 1e3:.0f} GB/s algorithmic  diff-vs-first {err:.1e}", flush=True)
    # forward + backward (grad_in1 and grad_W; in2 = harmonics of fixed positions needs none): the training use of the operator
    mod.kernel = 0
    xg = x.clone().requires_grad_(True)
    go = torch.randn(a.B, Dout, device=dev)

    def fb():
        for p in mod.parameters():
            p.grad = None
        xg.grad = None
        mod(xg, y).backward(go)

    fb(); fb()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(max(1, a.iters // 2)):
        fb()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / max(1, a.iters // 2)
    print(f"{ir} -> {a.out or ir}  B={a.B} forward + backward (grad_in1, grad_W): {ms:.3f} ms  "
          f"{a.B / ms / 1e6:.2f} Grows/s*1e-3", flush=True)
