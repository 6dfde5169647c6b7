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

import torch
from torch import nn

from . import ops
from .irreps import Irreps
from .l1_tensor_prod import L1TensorProduct
from .message import FusedMessage
from .radius_graph import RadiusGraph
from .tensor_product import SHTensorProduct


def _hidden_irreps(H: int, lmax: int):
    hid = Irreps(f"{H}x0e+{H}x1o" + (f"+{H}x2e" if lmax == 2 else ""))
    gated = Irreps(f"{H}x0e+{lmax * H}x0e+{H}x1o" + (f"+{H}x2e" if lmax == 2 else ""))
    return hid, gated


def _make_tp(in_irreps, out_irreps, lmax: int):
    """l_max = 1: the reference operator (pinned).  l_max = 2: its builder-defined generalisation -- also used with
    lmax_sh = 1 for the shapes the reference class rejects (it asserts lmax == 1 on both irreps, l1_tensor_prod.py:13-14,
    so a scalar-only head such as the 1x0e energy readout cannot be an L1TensorProduct)."""
    if lmax == 1:
        if Irreps(in_irreps).lmax == 1 and Irreps(out_irreps).lmax == 1:
            return L1TensorProduct(in_irreps, out_irreps)
        return SHTensorProduct(in_irreps, out_irreps, lmax_sh=1)
    return SHTensorProduct(in_irreps, out_irreps, lmax_sh=2)


class SEGNNLayer(nn.Module):
    _warned_unfused = False   # the fused -> unfused switch is announced once per process

    def __init__(self, H: int, lmax: int = 1):
        super().__init__()
        hid, gated = _hidden_irreps(H, lmax)
        self.H, self.lmax = H, lmax
        self.fused = True           # fused gather + TP + gate (+ segment-sum) kernels when the shapes allow it
        self.fuse_scatter = True    # segment-sum by fp32 atomics in the message kernel (sums agree to fp32 rounding, not
        #                             bit for bit); False = separate, bitwise reproducible e3_segment_sum
        self.fuse_message = True    # fp32: the whole message function in one launch (message.FusedMessage)
        self._msg = FusedMessage(lmax, H) if FusedMessage.supported(lmax, H) else None
        self.msg1 = _make_tp(hid + hid + Irreps("1x0e"), gated, lmax)
        self.msg2 = _make_tp(hid, gated, lmax)
        self.upd1 = _make_tp(hid + hid, gated, lmax)
        self.upd2 = _make_tp(hid, hid, lmax)

    def _gate(self, t):
        H = self.H
        return ops.gate(t, H, H) if self.lmax == 1 else ops.gate_blocks(t, H, [(1, H), (2, H)])

    def fused_available(self) -> bool:
        """Static part only (shapes with an MFMA instantiation); grad mode is looked at on every call.  True when the
        per-product fused kernels exist for BOTH message products and the update product."""
        f = getattr(self, "_fused_ok", None)
        if f is None:
            f = self._fused_ok = all(tp.fused_supported(True) for tp in (self.msg1, self.msg2, self.upd1))
        return f

    def fused_update_available(self) -> bool:
        """The update product alone (gather + concat + TP + gate in one launch) -- enough when the one-launch message kernel
        handles the messages (hidden 16 / 64: only the node-level products have per-product MFMA instantiations)."""
        f = getattr(self, "_fused_upd_ok", None)
        if f is None:
            f = self._fused_upd_ok = bool(self.upd1.fused_supported(True))
        return f

    def forward(self, h, g: RadiusGraph, Y, d, A, h_scale=None, halo=None, split=None):
        """-> (h_next, operand scale of h_next | None).  ``halo`` / ``split`` (sharding.GridHalo / SplitGraph): the layer
        refreshes the ghost rows of ``h`` itself -- in place -- and overlaps the transfer with the interior edges."""
        inference = not (torch.is_grad_enabled() and _needs_grad(self, h))
        if not inference and self.fused and not SEGNNLayer._warned_unfused:
            # the switch is visible: the fused MFMA kernels are inference kernels (no backward); a call that needs a gradient
            # runs the differentiable chain (gather -> TP -> gate -> TP -> gate -> segment-sum on the generic FMA kernels,
            # [E, width] tensors in HBM), which is several times slower -- tools/train_step_bench.py measures both
            SEGNNLayer._warned_unfused = True
            import warnings
            warnings.warn("SEGNNLayer: a gradient is required (grad mode on and an input or parameter requires grad) -> the "
                          "unfused differentiable chain runs instead of the fused MFMA inference kernels; wrap inference in "
                          "torch.no_grad() to get the fast path", RuntimeWarning, stacklevel=2)
        f32 = h.dtype == torch.float32
        r16 = self.fused and inference and self.fused_available()   # per-TP fused kernels (gather + TP + gate)
        # ---- message function -> aggregated messages a [N, width] ----
        one_launch = (self.fused and inference and self.fuse_message and self.fuse_scatter and self._msg is not None and
                      self._msg.supports(h.dtype))
        r16u = self.fused and inference and (r16 or (one_launch and self.fused_update_available()))   # update product
        if h.dtype == torch.bfloat16 and not ((one_launch or r16) and r16u):
            raise RuntimeError("bf16 storage needs the fused MFMA kernels (inference; hidden 32, or 64 at l_max = 2)")
        if halo is not None and not (one_launch and split is not None):
            halo.exchange(h)  # blocking refresh of the ghost rows, in place
        if one_launch and halo is not None and split is not None:
            # the refresh is posted first; interior edges (owned src) run while it is in flight, boundary edges after it
            # landed.  The operand scale has to be fixed BEFORE the refreshed ghost rows are known (the pre-mix table and
            # the interior launch use it): it is taken from the owned + stale ghost rows with NINE binades of head room
            # (joint maximum at 2^6 instead of 2^10) -- the fp16 (hi, lo) pair keeps an absolute error of 2^-25 of the scaled
            # range, so the lower target costs no accuracy, and refreshed rows up to 512 x larger than anything this rank
            # held stay inside the fp16 range.  Beyond that the overflow flag of the halo is raised (checked once per forward).
            tok = halo.start(h)
            if h_scale is None and f32:
                h_scale = ops.pow2_scale([h], target_log2=6)
            a, state = self._msg.forward(h, split.graph, self.msg1, self.msg2, h_scale, edges=split.interior,
                                         return_state=True)
            halo.finish(h, tok)
            if state is not None:
                # the pre-mix launch saw the STALE ghost rows: their row maxima (bound of a row's messages) are recomputed
                top = self._msg.refresh_row_max(state, h, halo.ghost_rows(), h_scale if f32 else None)
                if f32:
                    flag = top >= 2.0 ** 15
                    halo.overflow = flag if getattr(halo, "overflow", None) is None else (halo.overflow | flag)
            a = self._msg.forward(h, split.graph, self.msg1, self.msg2, h_scale, edges=split.boundary, cont=state).to(h.dtype)
            del state
        elif one_launch:
            # one launch: SH + TP #1 + gate + TP #2 + gate + segment-sum (message.FusedMessage)
            if h_scale is None and f32:
                h_scale = ops.pow2_scale([h])
            a = self._msg.forward(h, g, self.msg1, self.msg2, h_scale).to(h.dtype)
        elif r16:
            # gather + concat + TP + gate in one kernel each: no [E, 2D+1] / raw-TP tensors in HBM
            if d.dtype != h.dtype:
                d = d.to(h.dtype)
            m = self.msg1.forward_fused([(h, g.dst), (h, g.src), (d, None)], Y, gate=True,
                                        in_scale=ops.pow2_scale([h, d]) if f32 else None)
            a = None
            if self.fuse_scatter:  # segment-sum in the epilogue of message TP #2 where the library has that kernel
                a = self.msg2.forward_fused([(m, None)], Y, gate=True, scatter=(g.dst, g.rowptr.numel() - 1))
            if a is None:
                m = self.msg2.forward_fused([(m, None)], Y, gate=True)
                a = ops.segment_sum(m, g)
        else:
            m = ops.gather_concat(h, g, d)
            m = self._gate(self.msg1(m, Y))
            m = self._gate(self.msg2(m, Y))
            a = ops.segment_sum(m, g)
        # ---- node update ----
        if r16u:
            # operand scale of [h | a]: h's is known (the previous layer returned it), so only `a` is scanned
            sc = None
            if f32 and h_scale is not None and halo is None:
                sc = ops.join_pow2_scales(h_scale, ops.pow2_scale([a]))
            if self.upd2.fused_supported(False) and h.stride(-1) == 1:
                # the scale of u comes out of update #1's epilogue; update #2 adds the residual and emits the scale of the
                # new h in its own: no pass over [N, width] outside the two products
                u, u_scale = self.upd1.forward_fused([(h, None), (a, None)], A, gate=True, in_scale=sc,
                                                     out_scale=10 if f32 else None) if f32 else \
                    (self.upd1.forward_fused([(h, None), (a, None)], A, gate=True), None)
                if f32:
                    return self.upd2.forward_fused([(u, None)], A, gate=False, in_scale=u_scale, residual=h, out_scale=10)
                return self.upd2.forward_fused([(u, None)], A, gate=False, residual=h), None
            u = self.upd1.forward_fused([(h, None), (a, None)], A, gate=True, in_scale=sc)
        else:
            u = self._gate(self.upd1(torch.cat([h, a], 1), A))
        u = self.upd2(u, A)
        if f32 and inference:
            return ops.add_pow2_scale(h, u)
        return h + u, None


def _needs_grad(mod: nn.Module, *tensors) -> bool:
    return any(t.requires_grad for t in tensors) or any(p.requires_grad for p in mod.parameters())


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

    def forward(self, x: torch.Tensor, g: RadiusGraph, geometry=None, halo=None, split=None) -> torch.Tensor:
        """x [N, in_dim] node features in the graph's (Morton) order -> [N, out_dim] in the same order.

        ``halo`` (``sharding.SlabHalo``): when the cloud is spatially sharded, ghost rows of ``h`` are
        refreshed from their owners before every message-passing layer; only owned rows of the result
        are meaningful.  ``split`` (``halo.split_graph(g)``): edges into ghost rows dropped and the rest split into
        interior / boundary lists so that the refresh overlaps the interior edges."""
        if split is not None:
            g = split.graph
        if geometry is not None:
            Y, d, A = geometry
        else:
            # per-edge Y [E, (l_max+1)^2] and d [E] are only needed off the one-launch message path
            one_launch = (not (torch.is_grad_enabled() and _needs_grad(self, x)) and
                          all(l.fused and l.fuse_message and l.fuse_scatter and l._msg is not None and
                              l._msg.supports(x.dtype) for l in self.layers))
            Y, d, A = ops.edge_geometry(g, lmax=self.lmax, want_edge=not one_launch)
        if x.dtype == torch.bfloat16 and self.lmax != 2:
            raise RuntimeError("bf16 storage is implemented for l_max = 2 (BASELINE config 3)")
        h = self.embed(x, A)
        sc = None
        if halo is not None:
            halo.overflow = None
        for layer in self.layers:
            if halo is not None:
                sc = None  # the ghost rows are about to change: the scale of the previous layer's output is stale
            h, sc = layer(h, g, Y, d, A, sc, halo=halo, split=split)
        out = self.readout(h, A)
        if halo is not None and getattr(halo, "overflow", None) is not None and bool(halo.overflow):
            raise RuntimeError("sharded forward: a refreshed ghost row is more than 512 x larger than every row this rank held "
                               "when the layer's operand scale was fixed (fp16-split operands would overflow); run the layer "
                               "without `split` (blocking exchange) for such inputs")
        return out
