"""``SHTensorProduct`` — fully-connected tensor product of a feature (l <= 2) with the spherical
harmonics of an edge (lmax_sh in {1,2}); builder-defined generalisation of the reference's
``L1TensorProduct`` (which asserts lmax == 1, `L1TP.py:13-14`).  Contract: include/e3gnn.h
("General SH tensor product").  Same conventions as the reference where they overlap: one weight
matrix per output (l,p) class with rows in forward-concat order, ``U[-1,1]`` init, per-class norm
buffers ``sqrt((2l+1)/fan_in)`` ("component" x "element"), CG constants of `L1TP.py:91-94` for l <= 1.
Forward only (fp32 / fp64), ROCm tensors only, no CPU path.
"""
from __future__ import annotations

import ctypes
from math import sqrt

import torch
from torch import nn

from . import _lib, profiling
from .irreps import Irreps, as_blocks

CLASSES = ("l0e", "l0o", "l1e", "l1o", "l2e", "l2o")


class TPSegment(ctypes.Structure):
    _fields_ = [("base", ctypes.c_void_p), ("ld", ctypes.c_int64), ("row_index", ctypes.c_void_p),
                ("ncols", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class TPPlan:
    """``e3_tp_plan*`` handles (one per device) plus the packed-weight cache; shared by ``SHTensorProduct`` and by
    ``L1TensorProduct.forward_fused`` (the l <= 1 operator is the lmax_sh = 1 special case).  Deep-copy / pickle safe:
    only the irreps blocks are state."""

    def __init__(self, in1_irreps, out_irreps, lmax_sh):
        lib = _lib.load()
        self._blocks = ([tuple(b) for b in as_blocks(in1_irreps)], int(lmax_sh), [tuple(b) for b in as_blocks(out_irreps)])
        self._plans = _lib.DevicePlans("e3_tp_plan_create", "e3_tp_plan_destroy", *self._blocks)
        h = self._plans.handle(None)
        self.in1_dim = lib.e3_tp_in1_dim(h)
        self.in2_dim = lib.e3_tp_in2_dim(h)
        self.out_dim = lib.e3_tp_out_dim(h)
        self._packed = {}
        # algorithmic flops per row: sum over paths of 2 K M min(2 l1+1, 2 l3+1) (the contraction with W; the
        # per-row CG/feature algebra is excluded)
        n, M = {}, {}
        for l, p, mul in as_blocks(in1_irreps):
            n[(l, p)] = n.get((l, p), 0) + mul
        for l, p, mul in as_blocks(out_irreps):
            M[(l, p)] = M.get((l, p), 0) + mul
        fl = 0
        for (l3, p3), m3 in M.items():
            for (l1, p1), k in n.items():
                for l2 in range(lmax_sh + 1):
                    if abs(l1 - l2) <= l3 <= l1 + l2 and p1 * (-1) ** l2 == p3:
                        fl += 2 * k * m3 * min(2 * l1 + 1, 2 * l3 + 1)
        self.flops_per_row = fl

    def __deepcopy__(self, memo):
        return _rebuild_tpplan(*self._blocks)

    def __reduce__(self):
        return (_rebuild_tpplan, self._blocks)

    def handle(self, device=None):
        return self._plans.handle(device)

    def weight_shape(self, ci):
        rows, cols = ctypes.c_int(), ctypes.c_int()
        _lib.load().e3_tp_weight_shape(self.handle(), ci, ctypes.byref(rows), ctypes.byref(cols))
        return rows.value, cols.value

    def norm_len(self, ci):
        return _lib.load().e3_tp_norm_len(self.handle(), ci)

    def fused_supported(self, gate: bool) -> bool:
        return bool(_lib.load().e3_tp_fused_supported(self.handle(), 1 if gate else 0))

    def gated_width(self):
        """Width of the gated output for out irreps ``[H x0e | (nb H) x0e | H x1o (| H x2e)]`` (natural parity, H a
        multiple of 16): the nb H gate scalars disappear.  None when the out irreps do not have that shape."""
        blocks = self._blocks[2]
        if len(blocks) < 3 or any(b[0] != 0 or b[1] != 1 for b in blocks[:2]):
            return None
        H, gates = blocks[0][2], blocks[1][2]
        gated = blocks[2:]
        if H % 16 or gates != H * len(gated) or any(b[2] != H for b in gated):
            return None
        if [(b[0], b[1]) for b in gated] not in ([(1, -1)], [(1, -1), (2, 1)]):
            return None
        return H + sum((2 * b[0] + 1) * H for b in gated)

    def packed(self, ws, ns, dtype, device):
        """ws / ns: 6-entry lists (per class) of optional tensors.  The packed buffer is rebuilt when a parameter or
        buffer was replaced or modified through autograd-visible ops (``_version``); in-place edits through ``.data``
        do not bump the version counter -- call ``invalidate_packed()`` after those."""
        key = (dtype, device) + tuple((t.data_ptr(), t._version) if t is not None else None for t in ws + ns)
        hit = self._packed.get((dtype, device))
        stream = torch.cuda.current_stream(device)
        if hit is not None and hit[0] == key:
            if hit[2] != stream.cuda_stream:
                stream.wait_event(hit[3])  # packed on another stream: order this stream behind the pack kernels
            return hit[1]
        lib = _lib.load()
        code = _lib.dtype_code(dtype)
        for t in ws + ns:
            if t is not None and t.numel() and (t.dtype != dtype or t.device != device):
                raise RuntimeError(f"tensor product: parameter {t.dtype}/{t.device} vs input {dtype}/{device}")
        h = self.handle(device)
        nbytes = lib.e3_tp_packed_bytes(h, code)
        if nbytes < 0:
            raise RuntimeError(f"tensor product supports float32/float64/bfloat16, got {dtype}")
        packed = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        P6 = ctypes.c_void_p * 6
        ptr = lambda t: t.data_ptr() if (t is not None and t.numel()) else None
        wsc = [w.detach().contiguous() if w is not None else None for w in ws]
        nsc = [n.detach().contiguous() if n is not None else None for n in ns]
        _lib.check(lib.e3_tp_pack_weights(h, P6(*map(ptr, wsc)), P6(*map(ptr, nsc)), code,
                                          packed.data_ptr(), stream.cuda_stream), "e3_tp_pack_weights")
        ev = torch.cuda.Event()
        ev.record(stream)
        self._packed[(dtype, device)] = (key, packed, stream.cuda_stream, ev)
        return packed

    def invalidate_packed(self):
        self._packed = {}

    def forward_fused(self, ws, ns, segments, in2, gate: bool, tag="", scatter=None, in_scale=None, residual=None,
                      out_scale=None):
        """segments: [(tensor [R, ncols], row_index int32 [B] | None), ...]; -> [B, out_dim or gated width].
        scatter = (row_node int32 [B] ascending, n_nodes): rows are summed per node instead of stored (fused
        segment-sum, fp32 atomics) -> [n_nodes, width]; returns None when the library has no such kernel for this plan.
        in_scale: power-of-two operand scale of the segments (``ops.pow2_scale``); computed here when None (fp32).
        residual [B, width]: added to the result in the kernel's epilogue.  out_scale = target_log2 (int): also return the
        operand scale of the result, ``ops.pow2_scale([out], target_log2)``, taken in the same epilogue -> (out, scale)."""
        lib = _lib.load()
        B = in2.shape[0]
        dev = in2.device
        segs = (TPSegment * len(segments))()
        keep = []
        io = segments[0][0].dtype
        if io not in (torch.float32, torch.bfloat16):
            raise RuntimeError(f"forward_fused: float32 / bfloat16 storage only, got {io}")
        if in2.dtype != torch.float32:
            raise RuntimeError("forward_fused: in2 (spherical harmonics) is always float32")
        for i, (t, idx) in enumerate(segments):
            if t.dtype != io or not t.is_cuda:
                raise RuntimeError("forward_fused: all segments must share one dtype and live on a ROCm device")
            if t.dim() == 1:
                t = t.unsqueeze(1)
            if t.stride(-1) != 1:
                t = t.contiguous()
            keep.append(t)
            segs[i].base, segs[i].ld, segs[i].ncols = t.data_ptr(), t.stride(0), t.shape[1]
            if idx is not None:
                assert idx.dtype == torch.int32 and idx.numel() == B
                segs[i].row_index = idx.data_ptr()
            else:
                assert t.shape[0] == B
        width = self.out_dim
        if gate:
            width = self.gated_width()
            if width is None:
                raise RuntimeError("gate fusion needs out irreps [Hx0e | (blocks H)x0e | Hx1o (| Hx2e)], H a multiple of 16")
        if scatter is not None:
            row_node, n_nodes = scatter
            if not gate:
                return None
            assert row_node.dtype == torch.int32 and row_node.numel() == B and row_node.is_cuda
            out = torch.zeros((n_nodes, width), dtype=torch.float32, device=dev)  # fp32 sums for both storage types
        else:
            out = torch.empty((B, width), dtype=io, device=dev)
        if B == 0:
            if residual is not None:
                out = residual.clone()
            return (out, torch.tensor([1.0, 1.0, 0.0, 0.0], device=dev)) if out_scale is not None else out
        if in2.stride(-1) != 1:
            in2 = in2.contiguous()
        esz = out.element_size()
        with torch.cuda.device(dev):
            packed = self.packed(ws, ns, io, dev)
            if io == torch.float32 and in_scale is None:
                from . import ops
                in_scale = ops.pow2_scale(keep)
            sc = in_scale.data_ptr() if (in_scale is not None and io == torch.float32) else None
            stream = torch.cuda.current_stream(dev).cuda_stream
            t0 = profiling.begin() if profiling.enabled() else None
            h = self.handle(dev)
            if residual is not None or out_scale is not None:
                if scatter is not None:
                    raise RuntimeError("forward_fused: residual / out_scale do not combine with scatter")
                if residual is not None and (residual.shape != out.shape or residual.dtype != io or residual.stride(-1) != 1):
                    raise RuntimeError(f"forward_fused: residual must be {tuple(out.shape)} {io}, contiguous rows")
                sc4 = torch.empty(4, dtype=torch.float32, device=dev) if out_scale is not None else None
                _lib.check(lib.e3_tp_forward_fused_epilogue(
                    h, ctypes.byref(segs), len(segments), in2.data_ptr(), in2.stride(0), packed.data_ptr(), out.data_ptr(),
                    out.stride(0), B, _lib.dtype_code(io), 1 if gate else 0, sc,
                    residual.data_ptr() if residual is not None else None, residual.stride(0) if residual is not None else 0,
                    sc4.data_ptr() if sc4 is not None else None, int(out_scale) if out_scale is not None else 0, stream),
                    "e3_tp_forward_fused_epilogue")
            elif scatter is not None:
                st = lib.e3_tp_forward_fused_scatter(h, ctypes.byref(segs), len(segments), in2.data_ptr(),
                                                     in2.stride(0), packed.data_ptr(), scatter[0].data_ptr(),
                                                     out.data_ptr(), out.stride(0), B, _lib.dtype_code(io), 1, sc, stream)
                if st == 4:  # E3_ERR_UNSUPPORTED: no fused-scatter kernel for this plan / build
                    return None
                _lib.check(st, "e3_tp_forward_fused_scatter")
                if io != torch.float32:
                    out = out.to(io)  # one rounding of the fp32 sums (what e3_segment_sum_bf16 does)
            else:
                _lib.check(lib.e3_tp_forward_fused(h, ctypes.byref(segs), len(segments), in2.data_ptr(),
                                                   in2.stride(0), packed.data_ptr(), out.data_ptr(), out.stride(0), B,
                                                   _lib.dtype_code(io), 1 if gate else 0, sc, stream), "e3_tp_forward_fused")
            if t0 is not None:
                # algorithmic bytes: gathered segments count their SOURCE rows once (re-gathers are cache traffic)
                nb = sum((t.shape[0] * t.shape[1] * esz + (4 * B if idx is not None else 0)) for t, idx in
                         [(k, s[1]) for k, s in zip(keep, segments)]) + 4 * self.in2_dim * B + \
                     (esz * width * B if scatter is None else 4 * B + 4 * width * scatter[1])
                mode = "<bf16 storage, bf16 MFMA>" if io == torch.bfloat16 else "<fp16x3 split MFMA>"
                profiling.end(f"tp_fused{'+segsum' if scatter is not None else ''} {tag} B={B}", B, nb, t0, flops=self.flops_per_row * B,
                              kernel=(lib.e3_tp_last_fused_kernel() or b"e3::tp_fwd_mfma_r16_kernel").decode() + mode)
        if out_scale is not None:
            return out, sc4
        return out


def _rebuild_tpplan(in_blocks, lmax_sh, out_blocks):
    return TPPlan(in_blocks, out_blocks, lmax_sh)


# rows per pass of the GEMM-based backward: bounds the [rows, D3, K] feature / T workspace (bytes)
_BWD_WORKSPACE_BYTES = 1 << 30
# below this many rows the two generic kernels of e3_tp_backward are quicker than 2 + 2 x classes launches
_BWD_GEMM_MIN_ROWS = 512


_WGRAD_SLAB = 4096
# grad_W alone (no grad_in1 / grad_in2 wanted from this call) through e3_tp_backward_weights: features in LDS + fp32 MFMA, nothing
# of size [B, D3, K] in HBM (106 vs 134 ms for the 100 k-particle forward + backward).  False: always the operand pass + batched
# GEMM (tests/test_tp_backward_gpu.py compares the two)
_BWD_FUSED_WGRAD = True


def _wgrad_accumulate(gw, F, G):
    """gw [K, M] += F^T G with F [R, K], G [R, M], R >> K, M: the reduction runs over R, so one GEMM has K M / tile
    workgroups and a very long loop; slabs of ``_WGRAD_SLAB`` rows as one batched GEMM fill the device, the per-slab
    [K, M] results are summed afterwards."""
    R = F.shape[0]
    nb = R // _WGRAD_SLAB
    if nb >= 4:
        body = nb * _WGRAD_SLAB
        part = torch.bmm(F[:body].view(nb, _WGRAD_SLAB, -1).transpose(1, 2), G[:body].view(nb, _WGRAD_SLAB, -1))
        gw.add_(part.sum(0))
        if body < R:
            gw.addmm_(F[body:].t(), G[body:])
    else:
        gw.addmm_(F.t(), G)


def tp_backward(plan: "TPPlan", packed, in1, in2, grad_out, weights, need1, need2, need_w):
    """Gradients of ``e3_tp_forward`` -> (grad_in1 | None, grad_in2 | None (accumulation dtype), [grad_W per class | None]).

    ``weights`` / ``need_w``: 6-entry lists (per output class; None = class absent).  Large B runs as
    ``e3_tp_backward_operands`` -> per class ``grad_W += F^T G`` and ``T = G W^T`` (library GEMMs) ->
    ``e3_tp_backward_contract``, in row chunks that bound the workspace; small B on the generic kernels of
    ``e3_tp_backward``.  fp32 / fp64, the reference operator's gradients come from torch autograd over
    `l1_tensor_prod.py:240-299`."""
    lib = _lib.load()
    B, dev, dt = in1.shape[0], in1.device, in1.dtype
    acc = torch.float64 if dt == torch.float64 else torch.float32
    code = _lib.dtype_code(dt)
    bcast = in2.shape[0] == 1 and B != 1
    ld2 = 0 if bcast else in2.stride(0)
    g1 = torch.empty_like(in1, memory_format=torch.contiguous_format) if need1 else None
    g2 = (torch.zeros((1, plan.in2_dim), dtype=acc, device=dev) if bcast else
          torch.empty((B, plan.in2_dim), dtype=acc, device=dev)) if need2 else None
    gws = [torch.zeros(w.shape, dtype=acc, device=dev) if (w is not None and nw) else None
           for w, nw in zip(weights, need_w)]
    if B == 0:
        if g2 is not None:
            g2.zero_()
        return g1, g2, gws
    P6 = ctypes.c_void_p * 6
    ptr = lambda t: t.data_ptr() if t is not None else None
    h = plan.handle(dev)
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        if B < _BWD_GEMM_MIN_ROWS:
            _lib.check(lib.e3_tp_backward(h, in1.data_ptr(), in1.stride(0), in2.data_ptr(), ld2, packed.data_ptr(),
                                          grad_out.data_ptr(), grad_out.stride(0), ptr(g1), g1.stride(0) if need1 else 0,
                                          ptr(g2), 0 if (g2 is None or bcast) else g2.stride(0), P6(*map(ptr, gws)), B, code,
                                          stream), "e3_tp_backward")
            return g1, g2, gws
        if not need1 and not need2 and dt == torch.float32 and not bcast and _BWD_FUSED_WGRAD:
            # grad_W alone (the pinned operator's training case: grad_in1 comes from the transposed operator): features and
            # output gradient of a row tile built in LDS, contracted on the fp32 MFMA -- no [B, D3, K] operands in HBM
            st = lib.e3_tp_backward_weights(h, in1.data_ptr(), in1.stride(0), in2.data_ptr(), ld2, packed.data_ptr(),
                                            grad_out.data_ptr(), grad_out.stride(0), P6(*map(ptr, gws)), B, code, stream)
            if st == 0:
                return g1, g2, gws
            if st != 4:   # anything but E3_ERR_UNSUPPORTED (shape too large for the kernel: the GEMM path below)
                _lib.check(st, "e3_tp_backward_weights")
        shapes = [tuple(w.shape) if w is not None else None for w in weights]      # (K, M) per class
        per_row = sum((2 * (c >> 1) + 1) * (s[0] + s[1]) for c, s in enumerate(shapes) if s is not None)
        chunk = max(256, min(B, _BWD_WORKSPACE_BYTES // (per_row * (8 if dt == torch.float64 else 4))))
        work = torch.empty(chunk * per_row, dtype=acc, device=dev)
        need_rows = need1 or need2
        wacc = [w.detach().to(acc) if w is not None else None for w in weights]
        for r0 in range(0, B, chunk):
            r = min(chunk, B - r0)
            Fs, Gs, off = [None] * 6, [None] * 6, 0
            for c, s in enumerate(shapes):
                if s is None:
                    continue
                d3 = 2 * (c >> 1) + 1
                Fs[c] = work[off:off + r * d3 * s[0]].view(r * d3, s[0]); off += r * d3 * s[0]
                Gs[c] = work[off:off + r * d3 * s[1]].view(r * d3, s[1]); off += r * d3 * s[1]
            x, y, g = in1[r0:r0 + r], (in2 if bcast else in2[r0:r0 + r]), grad_out[r0:r0 + r]
            want_f = [Fs[c] if gws[c] is not None else None for c in range(6)]
            _lib.check(lib.e3_tp_backward_operands(h, x.data_ptr(), x.stride(0), y.data_ptr(), ld2, packed.data_ptr(),
                                                   g.data_ptr(), g.stride(0), P6(*map(ptr, want_f)), P6(*map(ptr, Gs)),
                                                   r, code, stream), "e3_tp_backward_operands")
            for c in range(6):
                if shapes[c] is None:
                    continue
                if gws[c] is not None:
                    _wgrad_accumulate(gws[c], Fs[c], Gs[c])
                if need_rows:
                    torch.mm(Gs[c], wacc[c].t(), out=Fs[c])     # T overwrites the features of this class
            if need_rows:
                g1c = g1[r0:r0 + r] if need1 else None
                g2c = (g2 if bcast else g2[r0:r0 + r]) if need2 else None
                _lib.check(lib.e3_tp_backward_contract(h, x.data_ptr(), x.stride(0), y.data_ptr(), ld2,
                                                       P6(*map(ptr, Fs)), ptr(g1c), g1c.stride(0) if need1 else 0,
                                                       ptr(g2c), 0 if (g2c is None or bcast) else g2c.stride(0), r, code,
                                                       stream), "e3_tp_backward_contract")
    return g1, g2, gws


class _SHTPFunction(torch.autograd.Function):
    """autograd bridge: forward = e3_tp_forward, backward = ``tp_backward`` (fp32 / fp64)."""

    @staticmethod
    def forward(ctx, mod, in1, in2, *weights):
        out = mod._forward_impl(in1, in2)
        ctx.mod = mod
        ctx.save_for_backward(in1, in2, *weights)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        mod = ctx.mod
        in1, in2, *weights = ctx.saved_tensors
        if grad_out.stride(-1) != 1 or grad_out.dtype != in1.dtype:
            grad_out = grad_out.to(in1.dtype).contiguous()
        present = [hasattr(mod, "weights_" + c) for c in CLASSES]
        it, ni = iter(weights), iter(ctx.needs_input_grad[3:])
        ws = [next(it) if pr else None for pr in present]
        nw = [next(ni) if pr else False for pr in present]
        with torch.cuda.device(in1.device):
            packed = mod._packed_weights(in1.dtype, in1.device)
        g1, g2, gws = tp_backward(mod._plan, packed, in1, in2, grad_out, ws, ctx.needs_input_grad[1],
                                  ctx.needs_input_grad[2], nw)
        out = [None, g1, g2.to(in2.dtype) if g2 is not None else None]
        out += [gw.to(w.dtype) if gw is not None else None for gw, w, pr in zip(gws, ws, present) if pr]
        return tuple(out)


class SHTensorProduct(nn.Module):
    def __init__(self, in1_irreps, out_irreps, lmax_sh: int = 2):
        super().__init__()
        self.iri1 = Irreps(in1_irreps) if isinstance(in1_irreps, str) else in1_irreps
        self.iro = Irreps(out_irreps) if isinstance(out_irreps, str) else out_irreps
        self.iri2 = Irreps.spherical_harmonics(lmax_sh)
        self.lmax_sh = lmax_sh
        self.exact = False  # True: fp32 inputs run on the generic fp32 FMA kernel instead of the fp16-split MFMA kernel
        self._plan = TPPlan(self.iri1, self.iro, lmax_sh)
        self.in1_dim, self.in2_dim, self.out_dim = self._plan.in1_dim, self._plan.in2_dim, self._plan.out_dim
        for ci, c in enumerate(CLASSES):
            rows, cols = self._plan.weight_shape(ci)
            nlen = self._plan.norm_len(ci)
            if rows > 0 and cols > 0:
                setattr(self, "weights_" + c, nn.Parameter(torch.rand((rows, cols)) * 2 - 1))
            l = ci >> 1
            val = sqrt((2 * l + 1) / rows) if rows > 0 else 1.0
            self.register_buffer("norm_" + c, torch.full((nlen,), val))


    def _tensors(self):
        ws = [getattr(self, "weights_" + c, None) for c in CLASSES]
        ns = [getattr(self, "norm_" + c) for c in CLASSES]
        return ws, ns

    def _packed_weights(self, dtype, device):
        ws, ns = self._tensors()
        return self._plan.packed(ws, ns, dtype, device)

    def fused_supported(self, gate: bool) -> bool:
        return self._plan.fused_supported(gate)

    def forward_fused(self, segments, in2, gate: bool = False, scatter=None, in_scale=None, residual=None, out_scale=None):
        """TP over ``in1 = [seg0[idx0] | seg1[idx1] | ...]`` (gather + concat fused), optional fused gate, optional
        fused segment-sum (``scatter=(row_node, n_nodes)``, see ``TPPlan.forward_fused``; None = unsupported), optional
        residual add and operand scale of the result (``residual``, ``out_scale``)."""
        ws, ns = self._tensors()
        return self._plan.forward_fused(ws, ns, segments, in2, gate, tag=f"{self.iri1}->{self.iro}", scatter=scatter,
                                        in_scale=in_scale, residual=residual, out_scale=out_scale)

    def forward(self, in1: torch.Tensor, in2: torch.Tensor) -> torch.Tensor:
        torch._assert(in1.shape[-1] == self.in1_dim,
                      f"Incorrect last dimension for in1 = {in1.shape[-1]}, required is {self.in1_dim}")
        torch._assert(in2.shape[-1] == self.in2_dim,
                      f"Incorrect last dimension for in2 = {in2.shape[-1]}, required is {self.in2_dim}")
        if not in1.is_cuda:
            raise RuntimeError("SHTensorProduct runs on ROCm tensors only; there is no CPU path")
        if torch.is_grad_enabled() and (in1.requires_grad or in2.requires_grad or
                                        any(p.requires_grad for p in self.parameters())):
            if in1.dtype not in (torch.float32, torch.float64):
                raise NotImplementedError("SHTensorProduct backward: float32 / float64 only")
            if in1.stride(-1) != 1:
                in1 = in1.contiguous()
            if in2.stride(-1) != 1:
                in2 = in2.contiguous()
            ws = [getattr(self, "weights_" + c) for c in CLASSES if hasattr(self, "weights_" + c)]
            return _SHTPFunction.apply(self, in1, in2, *ws)
        return self._forward_impl(in1, in2)

    def _forward_impl(self, in1: torch.Tensor, in2: torch.Tensor, in_scale=None) -> torch.Tensor:
        B = in1.shape[0]
        out = torch.empty((B, self.out_dim), dtype=in1.dtype, device=in1.device)
        if B == 0:
            return out
        want2 = torch.float32 if in1.dtype == torch.bfloat16 else in1.dtype  # bf16 storage keeps the SH in fp32
        if in2.dtype != want2:
            raise RuntimeError(f"SHTensorProduct: in2 must be {want2} for in1 {in1.dtype}, got {in2.dtype}")
        if in1.stride(-1) != 1:
            in1 = in1.contiguous()
        if in2.stride(-1) != 1:
            in2 = in2.contiguous()
        ld2 = 0 if (in2.shape[0] == 1 and B != 1) else in2.stride(0)
        lib = _lib.load()
        with torch.cuda.device(in1.device):
            packed = self._packed_weights(in1.dtype, in1.device)
            mfma = in1.dtype == torch.float32 and not self.exact and ld2 != 0 and self._plan.fused_supported(False)
            if mfma and in_scale is None:
                from . import ops
                in_scale = ops.pow2_scale([in1])
            stream = torch.cuda.current_stream(in1.device).cuda_stream
            t0 = profiling.begin() if profiling.enabled() else None
            _lib.check(lib.e3_tp_forward(self._plan.handle(in1.device), in1.data_ptr(), in1.stride(0), in2.data_ptr(), ld2,
                                         packed.data_ptr(), out.data_ptr(), out.stride(0), B,
                                         _lib.dtype_code(in1.dtype), in_scale.data_ptr() if (mfma and in_scale is not None) else None,
                                         1 if self.exact else 0, stream), "e3_tp_forward")
            if t0 is not None:
                profiling.end(f"tp_fwd(l<=2) {self.iri1}->{self.iro} B={B}", B,
                              in1.element_size() * (self.in1_dim + self.in2_dim + self.out_dim) * B, t0,
                              flops=self._plan.flops_per_row * B,
                              kernel="e3::tp_fwd_mfma_r16_kernel" if (in1.dtype != torch.float64 and not self.exact and
                                                                       self._plan.fused_supported(False)) else "e3::tp_fwd_generic_kernel")
        return out
