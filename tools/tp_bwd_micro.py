#!/usr/bin/env python3
"""One tensor product's backward at an edge-sized batch (development tool): message TP #1 of a hidden-32 layer,
B rows, forward + backward ITERS times.  LMAX=1|2, B=rows."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import models  # noqa
from scalable_e3_gnn_amd.tensor_product import SHTensorProduct

lmax = int(os.environ.get("LMAX", 1)); B = int(os.environ.get("B", 400_000)); iters = int(os.environ.get("ITERS", 3))
h = "32x0e+32x1o" + ("+32x2e" if lmax == 2 else "")
out = "32x0e+" + ("64x0e" if lmax == 2 else "32x0e") + "+32x1o" + ("+32x2e" if lmax == 2 else "")
mod = SHTensorProduct(f"{h}+{h}+1x0e", out, lmax).to("cuda:0")
x = torch.randn(B, mod.in1_dim, device="cuda:0", requires_grad=True)
y = torch.randn(B, mod.in2_dim, device="cuda:0", requires_grad=True)
g = torch.randn(B, mod.out_dim, device="cuda:0")
for i in range(iters + 1):
    if i == 1:
        torch.cuda.synchronize(); t0 = time.time()
    for p in mod.parameters():
        p.grad = None
    x.grad = y.grad = None
    mod(x, y).backward(g)
torch.cuda.synchronize()
print(f"lmax={lmax} B={B}: fwd+bwd {(time.time() - t0) / iters * 1e3:.2f} ms")
