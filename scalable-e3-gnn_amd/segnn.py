"""SEGNN forward built on ``L1TensorProduct`` (builder-defined architecture: the reference mount has
only the tensor product; this file is the repo's written contract for the rest).

Steerable E(3) message passing with l <= 1 (``Hx0e+Hx1o`` hidden features), for a radius graph given
as CSR-by-dst in Morton order (``radius_graph``):

    Y_e = SH_{l<=1}(x_src - x_dst)  (component norm.),  d_e = |x_src - x_dst|,  A_i = [1, mean_e Y1_e]
    h   = TP_embed(x ; A)
    per layer:
        m  = gate(TP_m1([h_dst | h_src | d] ; Y));   m = gate(TP_m2(m ; Y))
        a_i = sum_{e -> i} m_e
        u  = gate(TP_u1([h | a] ; A));               u = TP_u2(u ; A)
        h  = h + u
    out = TP_out(h ; A)

``gate``: TP output ``Hx0e + Hx0e + Hx1o`` = (scalars, gate scalars, vectors) -> ``silu(s)``, ``sigmoid(g) v``.
Every TP is the reference operator (`L1TensorProduct`), i.e. the pinned hot path.
"""
from __future__ import annotations

import torch
from torch import nn

from . import ops
from .irreps import Irreps
from .l1_tensor_prod import L1TensorProduct
from .radius_graph import RadiusGraph


class SEGNNLayer(nn.Module):
    def __init__(self, H: int):
        super().__init__()
        hid = Irreps(f"{H}x0e+{H}x1o")
        gated = Irreps(f"{H}x0e+{H}x0e+{H}x1o")
        self.H = H
        self.msg1 = L1TensorProduct(hid + hid + Irreps("1x0e"), gated)
        self.msg2 = L1TensorProduct(hid, gated)
        self.upd1 = L1TensorProduct(hid + hid, gated)
        self.upd2 = L1TensorProduct(hid, hid)

    def forward(self, h, g: RadiusGraph, Y, d, A):
        H = self.H
        m = ops.gather_concat(h, g, d)
        m = ops.gate(self.msg1(m, Y), H, H)
        m = ops.gate(self.msg2(m, Y), H, H)
        a = ops.segment_sum(m, g)
        u = ops.gate(self.upd1(torch.cat([h, a], 1), A), H, H)
        u = self.upd2(u, A)
        return h + u


class SEGNN(nn.Module):
    def __init__(self, in_irreps="1x0e+1x1o", hidden: int = 32, out_irreps="1x1o", num_layers: int = 4):
        super().__init__()
        self.hidden = hidden
        hid = Irreps(f"{hidden}x0e+{hidden}x1o")
        self.in_irreps, self.out_irreps = Irreps(in_irreps), Irreps(out_irreps)
        self.embed = L1TensorProduct(self.in_irreps, hid)
        self.layers = nn.ModuleList([SEGNNLayer(hidden) for _ in range(num_layers)])
        self.readout = L1TensorProduct(hid, self.out_irreps)

    def forward(self, x: torch.Tensor, g: RadiusGraph, geometry=None, halo=None) -> torch.Tensor:
        """x [N, in_dim] node features in the graph's (Morton) order -> [N, out_dim] in the same order.

        ``halo`` (``sharding.SlabHalo``): when the cloud is spatially sharded, ghost rows of ``h`` are
        refreshed from their owners before every message-passing layer; only owned rows of the result
        are meaningful."""
        Y, d, A = geometry if geometry is not None else ops.edge_geometry(g)
        h = self.embed(x, A)
        for layer in self.layers:
            if halo is not None:
                h = halo.exchange(h)
            h = layer(h, g, Y, d, A)
        return self.readout(h, A)
