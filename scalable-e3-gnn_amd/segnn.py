"""SEGNN forward built on ``L1TensorProduct`` (builder-defined architecture: the reference mount has
only the tensor product; this file is the repo's written contract for the rest).

Steerable E(3) message passing with ``l_max`` 1 (``Hx0e+Hx1o`` hidden features, every TP is the
reference operator) or 2 (``Hx0e+Hx1o+Hx2e``, TPs = ``SHTensorProduct``), for a radius graph given
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

import os

import torch
from torch import nn

from . import ops
from .irreps import Irreps
from .l1_tensor_prod import L1TensorProduct
from .radius_graph import RadiusGraph
from .tensor_product import SHTensorProduct


def _hidden_irreps(H: int, lmax: int):
    hid = Irreps(f"{H}x0e+{H}x1o" + (f"+{H}x2e" if lmax == 2 else ""))
    gated = Irreps(f"{H}x0e+{lmax * H}x0e+{H}x1o" + (f"+{H}x2e" if lmax == 2 else ""))
    return hid, gated


def _make_tp(in_irreps, out_irreps, lmax: int):
    """l_max = 1: the reference operator (pinned).  l_max = 2: its builder-defined generalisation."""
    if lmax == 1:
        return L1TensorProduct(in_irreps, out_irreps)
    return SHTensorProduct(in_irreps, out_irreps, lmax_sh=2)


class SEGNNLayer(nn.Module):
    def __init__(self, H: int, lmax: int = 1):
        super().__init__()
        hid, gated = _hidden_irreps(H, lmax)
        self.H, self.lmax = H, lmax
        self.fused = True  # use the fused gather+TP+gate kernel when the shapes allow it
        # fused segment-sum (atomics: sums agree to fp32 rounding, not bit for bit); E3_FUSED_SCATTER=0 disables
        self.fuse_scatter = os.environ.get("E3_FUSED_SCATTER", "1") != "0"
        self.msg1 = _make_tp(hid + hid + Irreps("1x0e"), gated, lmax)
        self.msg2 = _make_tp(hid, gated, lmax)
        self.upd1 = _make_tp(hid + hid, gated, lmax)
        self.upd2 = _make_tp(hid, hid, lmax)

    def _gate(self, t):
        H = self.H
        return ops.gate(t, H, H) if self.lmax == 1 else ops.gate_blocks(t, H, [(1, H), (2, H)])

    def _fused(self) -> bool:
        f = getattr(self, "_fused_ok", None)
        if f is None:
            f = (self.H == 32 and not torch.is_grad_enabled() and
                 all(tp.fused_supported(True) for tp in (self.msg1, self.msg2, self.upd1)))
            self._fused_ok = f
        return f and not torch.is_grad_enabled()

    def forward(self, h, g: RadiusGraph, Y, d, A):
        if h.dtype == torch.bfloat16 and not (self.fused and self._fused()):
            raise RuntimeError("bf16 storage needs the fused MFMA path (H = 32, torch.no_grad())")
        if self.fused and self._fused():
            # gather + concat + TP + gate in one kernel each: no [E, 2D+1] / raw-TP tensors in HBM
            if d.dtype != h.dtype:
                d = d.to(h.dtype)
            m = self.msg1.forward_fused([(h, g.dst), (h, g.src), (d, None)], Y, gate=True)
            # message TP #2 with the segment-sum fused into its epilogue where the library has that kernel (l_max = 2): the [E, width] messages are never written; otherwise two kernels
            a = None
            if self.fuse_scatter:
                a = self.msg2.forward_fused([(m, None)], Y, gate=True, scatter=(g.dst, g.rowptr.numel() - 1))
            if a is None:
                m = self.msg2.forward_fused([(m, None)], Y, gate=True)
                a = ops.segment_sum(m, g)
            u = self.upd1.forward_fused([(h, None), (a, None)], A, gate=True)
            u = self.upd2(u, A)
            return h + u
        m = ops.gather_concat(h, g, d)
        m = self._gate(self.msg1(m, Y))
        m = self._gate(self.msg2(m, Y))
        a = ops.segment_sum(m, g)
        u = self._gate(self.upd1(torch.cat([h, a], 1), A))
        u = self.upd2(u, A)
        return h + u


class SEGNN(nn.Module):
    def __init__(self, in_irreps="1x0e+1x1o", hidden: int = 32, out_irreps="1x1o", num_layers: int = 4, lmax: int = 1):
        super().__init__()
        assert lmax in (1, 2)
        self.hidden, self.lmax = hidden, lmax
        hid, _ = _hidden_irreps(hidden, lmax)
        self.in_irreps, self.out_irreps = Irreps(in_irreps), Irreps(out_irreps)
        self.embed = _make_tp(self.in_irreps, hid, lmax)
        self.layers = nn.ModuleList([SEGNNLayer(hidden, lmax) for _ in range(num_layers)])
        self.readout = _make_tp(hid, self.out_irreps, lmax)

    def forward(self, x: torch.Tensor, g: RadiusGraph, geometry=None, halo=None) -> torch.Tensor:
        """x [N, in_dim] node features in the graph's (Morton) order -> [N, out_dim] in the same order.

        ``halo`` (``sharding.SlabHalo``): when the cloud is spatially sharded, ghost rows of ``h`` are
        refreshed from their owners before every message-passing layer; only owned rows of the result
        are meaningful."""
        Y, d, A = geometry if geometry is not None else ops.edge_geometry(g, lmax=self.lmax)
        if x.dtype == torch.bfloat16 and self.lmax != 2:
            raise RuntimeError("bf16 storage is implemented for l_max = 2 (BASELINE config 3)")
        h = self.embed(x, A)
        for layer in self.layers:
            if halo is not None:
                h = halo.exchange(h)
            h = layer(h, g, Y, d, A)
        return self.readout(h, A)
