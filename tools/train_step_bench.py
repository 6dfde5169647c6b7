#!/usr/bin/env python3
"""Forward + backward timings (development / profiles tool).

  (a) BASELINE.json configs[3]: 128 synthetic QM9-shaped molecules (3..29 atoms, positions randn * 1.5 A, r = 5 A), l_max = 2,
      4 layers, H = 32: energies + forces (-dE/dpos) + dE/dparam = one training step on energies with forces predicted;
  (b) BASELINE.json configs[1]: 100 k particles, l_max = 1, 4 layers, H = 32: forward + backward of sum(out^2) w.r.t. every
      parameter.
The one-launch fused kernels are inference kernels; a call that needs a gradient runs the differentiable chain (SEGNNLayer
warns once and names the reason): per-product forward kernels (MFMA where the shape has one) with [E, width] tensors
materialised, and the operands -> library GEMMs -> contract backward of tensor_product.tp_backward.  HIP-event times, ms per step after a warm-up.
"""
import json, math, os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import models  # noqa
from scalable_e3_gnn_amd.batched import BatchedEnergyModel
from scalable_e3_gnn_amd.radius_graph import radius_graph
from scalable_e3_gnn_amd.segnn import SEGNN

dev = "cuda:0"
iters = int(os.environ.get("ITERS", 5))


def timed(fn):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


out = {}
warnings.simplefilter("always")
# ---- (a) 128 molecules, l_max = 2, energy + force head ----
g = torch.Generator().manual_seed(3)
counts = torch.randint(3, 30, (128,), generator=g)
batch = torch.repeat_interleave(torch.arange(128), counts)
N = int(counts.sum())
pos = (torch.randn(N, 3, generator=g) * 1.5).to(dev)
x = torch.randn(N, 4, generator=g).to(dev)
batch = batch.to(dev)
torch.manual_seed(0)
model = BatchedEnergyModel("1x0e+1x1o", 32, 4, lmax=2).to(dev).train()
params = [p for p in model.parameters() if p.requires_grad]


def step_a():
    for p in params:
        p.grad = None
    e, f = model(x, pos, batch, 5.0, forces=True)
    e.sum().backward()
    return e, f


def fwd_a():
    with torch.no_grad():
        return model(x, pos, batch, 5.0)


model.eval(); t_inf = timed(fwd_a); model.train()
t_train = timed(step_a)
out["qm9_128mol_lmax2"] = {"atoms": N, "forward_inference_ms": t_inf, "energy_forces_param_grads_ms": t_train,
                           "note": "inference = fused MFMA kernels; training step = differentiable chain, backward = operands -> GEMMs -> "
                                   "contract (forward with autograd graph, -dE/dpos, dE/dparam)"}
print(json.dumps(out["qm9_128mol_lmax2"]), flush=True)

# ---- (b) 100 k particles, l_max = 1 ----
n = 100_000
pos = torch.rand(n, 3, device=dev, generator=torch.Generator(device=dev).manual_seed(1234))
xb = torch.randn(n, 4, device=dev)
r = float((3.0 * 24.0 / (4.0 * math.pi * n)) ** (1.0 / 3.0))
torch.manual_seed(0)
m1 = SEGNN("1x0e+1x1o", 32, "1x1o", 4, lmax=1).to(dev)
p1 = [p for p in m1.parameters() if p.requires_grad]
gr = radius_graph(pos, r, [0, 0, 0], [1, 1, 1])
xs = xb[gr.perm.long()]


def step_b():
    for p in p1:
        p.grad = None
    o = m1(xs, gr)
    o.square().sum().backward()


def fwd_b():
    with torch.no_grad():
        return m1(xs, gr)


t_inf = timed(fwd_b)
t_train = timed(step_b)
out["particles_100k_lmax1"] = {"particles": n, "edges": gr.num_edges, "forward_inference_ms": t_inf,
                               "forward_backward_ms": t_train,
                               "note": "graph build excluded; inference = fused MFMA kernels; forward + backward = differentiable "
                                       "chain ([E, width] tensors materialised), backward = operands -> GEMMs -> contract"}
print(json.dumps(out["particles_100k_lmax1"]), flush=True)
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
