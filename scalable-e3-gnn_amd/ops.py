"""Edge / node stages of the SEGNN forward (host side of ``e3_edge_geometry``, ``e3_gather_concat``,
``e3_gate``, ``e3_segment_sum`` and their ``*_backward`` in include/e3gnn.h).  Builder-defined,
fp32.  Every op is differentiable (torch.autograd.Function over the HIP backward kernels): with the tensor products'
own backward a whole SEGNN layer has parameter gradients and forces.  ROCm tensors only; no CPU path."""
from __future__ import annotations

import torch

from . import _lib
from .radius_graph import RadiusGraph


def _check(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise RuntimeError(f"{name}: ROCm tensor required (no CPU path)")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name}: float32 required, got {t.dtype}")


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _wants_grad(*ts) -> bool:
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in ts)


def _edge_geometry_raw(pos4, g, want_dist, want_node_attr, lmax, want_edge=True):
    N, E, dev = g.rowptr.numel() - 1, g.num_edges, pos4.device
    ny = (lmax + 1) ** 2
    Y = torch.empty((E, ny), dtype=torch.float32, device=dev) if want_edge else None
    d = torch.empty(E, dtype=torch.float32, device=dev) if (want_dist and want_edge) else None
    A = torch.empty((N, ny), dtype=torch.float32, device=dev) if want_node_attr else None
    fn = {1: "e3_edge_geometry", 2: "e3_edge_geometry_l2"}[lmax]
    with torch.cuda.device(dev):
        _lib.check(getattr(_lib.load(), fn)(pos4.data_ptr(), g.rowptr.data_ptr(), g.src.data_ptr(), N,
                                                Y.data_ptr() if Y is not None else None,
                                                d.data_ptr() if d is not None else None,
                                                A.data_ptr() if A is not None else None, _stream(pos4)),
                   "e3_edge_geometry")
    return Y, d, A


class _EdgeGeometryFn(torch.autograd.Function):
    """pos [N,3] (graph order) -> Y, d, A; backward = e3_edge_geometry_backward (dY/dpos, dd/dpos, dA/dpos)."""

    @staticmethod
    def forward(ctx, pos, g, lmax):
        pos4 = torch.zeros((pos.shape[0], 4), dtype=torch.float32, device=pos.device)
        pos4[:, :3] = pos
        Y, d, A = _edge_geometry_raw(pos4, g, True, True, lmax)
        ctx.g, ctx.lmax = g, lmax
        ctx.save_for_backward(pos4)
        return Y, d, A

    @staticmethod
    def backward(ctx, gY, gd, gA):
        (pos4,) = ctx.saved_tensors
        g = ctx.g
        N = pos4.shape[0]
        gpos = torch.empty((N, 3), dtype=torch.float32, device=pos4.device)
        c = lambda t: t.contiguous() if t is not None else None
        gY, gd, gA = c(gY), c(gd), c(gA)
        p = lambda t: t.data_ptr() if t is not None else None
        with torch.cuda.device(pos4.device):
            _lib.check(_lib.load().e3_edge_geometry_backward(pos4.data_ptr(), g.rowptr.data_ptr(), g.src.data_ptr(), N,
                                                             ctx.lmax, p(gY), p(gd), p(gA), gpos.data_ptr(),
                                                             _stream(pos4)), "e3_edge_geometry_backward")
        return gpos, None, None


def edge_geometry(g: RadiusGraph, want_dist=True, want_node_attr=True, lmax: int = 1, pos: torch.Tensor | None = None,
                  want_edge=True):
    """-> Y [E,(lmax+1)^2] | None, d [E] | None, A [N,(lmax+1)^2] | None.  ``want_edge=False``: only the node attribute
    (the fused message kernel computes the spherical harmonics of its edges itself).

    ``pos`` [N,3] (graph order, i.e. ``original_pos[g.perm]``): when given and it requires grad, the three outputs are
    differentiable w.r.t. it (forces = -dE/dpos); otherwise the graph's own ``pos4`` is used."""
    _check(g.pos4, "pos4")
    if pos is not None and _wants_grad(pos):
        _check(pos, "pos")
        return _EdgeGeometryFn.apply(pos, g, lmax)
    pos4 = g.pos4
    if pos is not None:
        pos4 = torch.zeros_like(g.pos4)
        pos4[:, :3] = pos
    return _edge_geometry_raw(pos4, g, want_dist, want_node_attr, lmax, want_edge)


class _GatherConcatFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, extra, g):
        ctx.g, ctx.has_extra = g, extra is not None
        ctx.D, ctx.nx = h.shape[1], (extra.reshape(g.num_edges, -1).shape[1] if extra is not None else 0)
        return _gather_concat_raw(h, g, extra)

    @staticmethod
    def backward(ctx, gm):
        g, D, nx = ctx.g, ctx.D, ctx.nx
        gm = gm.contiguous()
        N = g.rowptr.numel() - 1
        gh = torch.empty((N, D), dtype=torch.float32, device=gm.device)
        gx = torch.empty((g.num_edges, nx), dtype=torch.float32, device=gm.device) if nx else None
        with torch.cuda.device(gm.device):
            _lib.check(_lib.load().e3_gather_concat_backward(gm.data_ptr(), gm.stride(0), D, g.rowptr.data_ptr(),
                                                             g.src.data_ptr(), N, nx, gh.data_ptr(), gh.stride(0),
                                                             gx.data_ptr() if gx is not None else None, _stream(gm)),
                       "e3_gather_concat_backward")
        if gx is not None and nx == 1:
            gx = gx  # [E,1]; reshaped to the caller's shape by autograd below
        return gh, gx, None


def _gather_concat_raw(h, g, extra):
    if h.stride(-1) != 1:
        h = h.contiguous()
    N, D = h.shape
    E = g.num_edges
    nx = 0
    if extra is not None:
        extra = extra.reshape(E, -1).contiguous()
        nx = extra.shape[1]
    out = torch.empty((E, 2 * D + nx), dtype=torch.float32, device=h.device)
    with torch.cuda.device(h.device):
        _lib.check(_lib.load().e3_gather_concat(h.data_ptr(), h.stride(0), D, g.rowptr.data_ptr(), g.src.data_ptr(), N,
                                                extra.data_ptr() if nx else None, nx, out.data_ptr(), out.stride(0),
                                                _stream(h)), "e3_gather_concat")
    return out


def gather_concat(h: torch.Tensor, g: RadiusGraph, extra: torch.Tensor | None = None) -> torch.Tensor:
    """[E, 2D+n_extra] = [h[dst] | h[src] | extra]  (differentiable w.r.t. h and extra)"""
    _check(h, "h")
    if extra is not None:
        _check(extra, "extra")
    if _wants_grad(h, extra):
        ex2 = extra.reshape(g.num_edges, -1) if extra is not None else None
        return _GatherConcatFn.apply(h, ex2, g)
    return _gather_concat_raw(h, g, extra)


class _GateBlocksFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ns, blocks):
        ctx.ns, ctx.blocks = ns, blocks
        ctx.save_for_backward(x)
        return _gate_blocks_raw(x, ns, blocks)

    @staticmethod
    def backward(ctx, go):
        import ctypes
        (x,) = ctx.saved_tensors
        go = go.contiguous()
        gi = torch.empty_like(x)
        blocks = ctx.blocks
        ls = (ctypes.c_int32 * len(blocks))(*[l for l, _ in blocks])
        ms = (ctypes.c_int32 * len(blocks))(*[m for _, m in blocks])
        with torch.cuda.device(x.device):
            _lib.check(_lib.load().e3_gate_blocks_backward(x.data_ptr(), x.stride(0), go.data_ptr(), go.stride(0),
                                                           gi.data_ptr(), gi.stride(0), x.shape[0], ctx.ns, len(blocks),
                                                           ls, ms, _stream(x)), "e3_gate_blocks_backward")
        return gi, None, None


def gate(x: torch.Tensor, ns: int, nv: int) -> torch.Tensor:
    """[B, ns + nv + 3nv] (scalars | gates | vectors) -> [B, ns + 3nv] = [silu(s) | sigmoid(g) v]  (differentiable)"""
    _check(x, "x")
    if x.stride(-1) != 1:
        x = x.contiguous()
    B = x.shape[0]
    assert x.shape[1] == ns + 4 * nv, (x.shape, ns, nv)
    if _wants_grad(x):
        return _GateBlocksFn.apply(x, ns, ((1, nv),))
    out = torch.empty((B, ns + 3 * nv), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().e3_gate(x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), B, ns, nv,
                                       _stream(x)), "e3_gate")
    return out


class _SegmentSumFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, msg, g):
        ctx.g = g
        return _segment_sum_raw(msg, g)

    @staticmethod
    def backward(ctx, ga):
        g = ctx.g
        ga = ga.contiguous()
        N, D = ga.shape
        gm = torch.empty((g.num_edges, D), dtype=torch.float32, device=ga.device)
        with torch.cuda.device(ga.device):
            _lib.check(_lib.load().e3_segment_sum_backward(ga.data_ptr(), ga.stride(0), g.rowptr.data_ptr(), N, D,
                                                           gm.data_ptr(), max(gm.stride(0), D), _stream(ga)),
                       "e3_segment_sum_backward")
        return gm, None


def _segment_sum_raw(msg, g):
    if msg.stride(-1) != 1:
        msg = msg.contiguous()
    N = g.rowptr.numel() - 1
    D = msg.shape[1]
    agg = torch.empty((N, D), dtype=msg.dtype, device=msg.device)
    fn = "e3_segment_sum" if msg.dtype == torch.float32 else "e3_segment_sum_bf16"
    with torch.cuda.device(msg.device):
        _lib.check(getattr(_lib.load(), fn)(msg.data_ptr(), msg.stride(0), g.rowptr.data_ptr(), N, D,
                                            agg.data_ptr(), agg.stride(0), _stream(msg)), fn)
    return agg


def segment_sum(msg: torch.Tensor, g: RadiusGraph) -> torch.Tensor:
    """agg[i] = sum of msg rows of CSR row i (fixed order, reproducible); fp32 (differentiable), or bf16 storage with fp32
    accumulation"""
    if not msg.is_cuda:
        raise RuntimeError("msg: ROCm tensor required (no CPU path)")
    if msg.dtype not in (torch.float32, torch.bfloat16):
        raise RuntimeError(f"segment_sum: float32 / bfloat16 required, got {msg.dtype}")
    if msg.dtype == torch.float32 and _wants_grad(msg):
        return _SegmentSumFn.apply(msg, g)
    return _segment_sum_raw(msg, g)


def _gate_blocks_raw(x, ns, blocks):
    import ctypes
    B = x.shape[0]
    wide = sum(m * (2 * l + 1) for l, m in blocks)
    out = torch.empty((B, ns + wide), dtype=torch.float32, device=x.device)
    ls = (ctypes.c_int32 * len(blocks))(*[l for l, _ in blocks])
    ms = (ctypes.c_int32 * len(blocks))(*[m for _, m in blocks])
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().e3_gate_blocks(x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), B, ns,
                                              len(blocks), ls, ms, _stream(x)), "e3_gate_blocks")
    return out


def gate_blocks(x: torch.Tensor, ns: int, blocks) -> torch.Tensor:
    """x = [ns scalars | one gate per gated channel | gated blocks]; blocks = [(l, mul), ...]
    -> [silu(scalars) | sigmoid(gate) * block]  (differentiable)"""
    _check(x, "x")
    if x.stride(-1) != 1:
        x = x.contiguous()
    blocks = tuple((int(l), int(m)) for l, m in blocks)
    ng = sum(m for _, m in blocks)
    wide = sum(m * (2 * l + 1) for l, m in blocks)
    assert x.shape[1] == ns + ng + wide, (x.shape, ns, blocks)
    if _wants_grad(x):
        return _GateBlocksFn.apply(x, ns, blocks)
    return _gate_blocks_raw(x, ns, blocks)


def pow2_scale(tensors, target_log2: int = 10) -> torch.Tensor:
    """Device-resident operand scale ``[s, 1/s, scratch, scratch]`` (fp32) of up to 4 fp32 tensors: ``s`` is the power of
    two that puts their joint max |x| at ``2^target_log2`` (``e3_pow2_scale``; no host sync).  The fp16-split MFMA
    kernels take it as ``in_scale``."""
    import ctypes
    from .tensor_product import TPSegment
    ts = []
    for t in tensors:
        _check(t, "pow2_scale")
        if t.dim() == 1:
            t = t.unsqueeze(1)
        if t.stride(-1) != 1:
            t = t.contiguous()
        ts.append(t)
    assert 1 <= len(ts) <= 4
    dev = ts[0].device
    out = torch.empty(4, dtype=torch.float32, device=dev)
    segs = (TPSegment * len(ts))()
    rows = (ctypes.c_int64 * len(ts))()
    for i, t in enumerate(ts):
        segs[i].base, segs[i].ld, segs[i].ncols = t.data_ptr(), t.stride(0), t.shape[1]
        rows[i] = t.shape[0]
    with torch.cuda.device(dev):
        _lib.check(_lib.load().e3_pow2_scale(ctypes.byref(segs), rows, len(ts), target_log2, out.data_ptr(),
                                             _stream(out)), "e3_pow2_scale")
    return out


def join_pow2_scales(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """The operand scale of the union of two tensor sets from their scales (same target): the smaller ``s``, the larger
    ``1/s`` -- lets a caller that already knows the scale of one operand (``h``: returned by ``add_pow2_scale``) scan
    only the other one.  Device side, no host sync."""
    return torch.stack([torch.minimum(a[0], b[0]), torch.maximum(a[1], b[1]), torch.maximum(a[2], b[2]), a[3]])


def add_pow2_scale(h: torch.Tensor, u: torch.Tensor, target_log2: int = 10):
    """``h + u`` and the operand scale of the sum in one pass (``e3_add_pow2_scale``) -> (sum, scale)."""
    _check(h, "h")
    _check(u, "u")
    assert h.shape == u.shape
    h, u = h.contiguous(), u.contiguous()
    out = torch.empty_like(h)
    sc = torch.empty(4, dtype=torch.float32, device=h.device)
    if h.numel() % 4 or (h.data_ptr() | u.data_ptr() | out.data_ptr()) % 16:
        torch.add(h, u, out=out)
        return out, pow2_scale([out], target_log2)
    with torch.cuda.device(h.device):
        _lib.check(_lib.load().e3_add_pow2_scale(h.data_ptr(), u.data_ptr(), out.data_ptr(), h.numel(), target_log2,
                                                 sc.data_ptr(), _stream(h)), "e3_add_pow2_scale")
    return out, sc
