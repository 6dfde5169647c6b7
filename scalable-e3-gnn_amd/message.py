"""Host side of the fused SEGNN message function (``e3_msg_*`` in include/e3gnn.h; kernel: csrc/e3_msg_fused.hip).

    a_i = sum_{e -> i} gate(TP_m2(gate(TP_m1([h_dst | h_src | d_e]; Y_e)); Y_e))

One launch per layer (plus a per-node pre-mix launch): spherical harmonics, both message products, both gates and the
segment-sum over ``dst``; nothing of size ``E x width`` reaches HBM.  Builder-defined; the two
tensor products are the ``e3_tp_*`` operators, i.e. the reference operator for l <= 1.  fp32 storage, ROCm tensors only.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib, profiling
from .radius_graph import RadiusGraph

_NAT = ("l0e", "l1o", "l2e")  # natural-parity classes = output degrees 0, 1, 2


class FusedMessage:
    """Plan + packed weights of the fused message kernel for one (l_max, hidden) pair.  Deep-copy / pickle safe."""

    def __init__(self, lmax: int, hidden: int):
        self.lmax, self.hidden = int(lmax), int(hidden)
        self._plans = _lib.DevicePlans("e3_msg_plan_create", "e3_msg_plan_destroy", self.lmax, self.hidden)
        self._packed = {}
        self.tiles_per_block = 0  # 0 = library default

    def __deepcopy__(self, memo):
        return FusedMessage(self.lmax, self.hidden)

    def __reduce__(self):
        return (FusedMessage, (self.lmax, self.hidden))

    @staticmethod
    def supported(lmax: int, hidden: int) -> bool:
        return lmax in (1, 2) and hidden in (16, 32, 64)

    @property
    def width(self) -> int:
        return self.hidden * (self.lmax + 1) ** 2

    def _tensors(self, tp):
        ws = [getattr(tp, "weights_" + c, None) for c in _NAT]
        ns = [getattr(tp, "norm_" + c, None) for c in _NAT]
        return ws, ns

    def supports(self, dtype) -> bool:
        if dtype == torch.float32:
            return True
        return dtype == torch.bfloat16 and bool(_lib.load().e3_msg_supports(self._plans.handle(None), _lib.E3_BF16))

    def packed(self, msg1, msg2, device, dtype=torch.float32) -> torch.Tensor:
        lib = _lib.load()
        w1, n1 = self._tensors(msg1)
        w2, n2 = self._tensors(msg2)
        ts = w1 + n1 + w2 + n2
        key = tuple((t.data_ptr(), t._version) if t is not None else None for t in ts)
        hit = self._packed.get((device, dtype))
        stream = torch.cuda.current_stream(device)
        if hit is not None and hit[0] == key:
            if hit[2] != stream.cuda_stream:
                stream.wait_event(hit[3])
            return hit[1]
        h = self._plans.handle(device)
        for tpi, ws in ((1, w1), (2, w2)):
            for l in range(self.lmax + 1):
                rows, cols = ctypes.c_int(), ctypes.c_int()
                lib.e3_msg_weight_shape(h, tpi, l, ctypes.byref(rows), ctypes.byref(cols))
                w = ws[l]
                if w is None or tuple(w.shape) != (rows.value, cols.value) or w.dtype != dtype or w.device != device:
                    raise RuntimeError(f"fused message: TP #{tpi} weights of degree {l} must be {dtype} "
                                       f"[{rows.value}, {cols.value}] on {device}, got "
                                       f"{None if w is None else (tuple(w.shape), w.dtype, w.device)}")
        P3 = ctypes.c_void_p * 3
        keep = [t.detach().contiguous() if t is not None else None for t in ts]
        ptr = lambda t: t.data_ptr() if (t is not None and t.numel()) else None
        packed = torch.empty(int(lib.e3_msg_packed_bytes(h)), dtype=torch.uint8, device=device)
        with torch.cuda.device(device):
            for t in keep[3:6] + keep[9:12]:
                if t is not None and t.numel() and t.dtype != dtype:
                    raise RuntimeError(f"fused message: norm buffers must be {dtype} (cast the module), got {t.dtype}")
            _lib.check(lib.e3_msg_pack_weights(h, P3(*map(ptr, keep[0:3])), P3(*map(ptr, keep[3:6])),
                                               P3(*map(ptr, keep[6:9])), P3(*map(ptr, keep[9:12])), _lib.dtype_code(dtype),
                                               packed.data_ptr(), stream.cuda_stream), "e3_msg_pack_weights")
        ev = torch.cuda.Event()
        ev.record(stream)
        self._packed[(device, dtype)] = (key, packed, stream.cuda_stream, ev)
        return packed

    def flops_per_edge(self) -> int:
        """Algorithmic flops of the two message products per edge: sum over paths of 2 K M min(2 l1+1, 2 l3+1)."""
        H, L = self.hidden, self.lmax
        fl = 0
        for tp, n in ((1, [2 * H + 1, 2 * H, 2 * H]), (2, [H, H, H])):
            for l3 in range(L + 1):
                M = H * (1 + L) if l3 == 0 else H
                for l1 in range(L + 1):
                    for l2 in range(L + 1):
                        if abs(l1 - l2) <= l3 <= l1 + l2 and (l1 + l2 + l3) % 2 == 0:
                            fl += 2 * n[l1] * M * min(2 * l1 + 1, 2 * l3 + 1)
        return fl

    def refresh_row_max(self, state, h: torch.Tensor, rows: torch.Tensor, in_scale: torch.Tensor | None):
        """The pre-mix launch leaves max |h[n] in_scale| per node behind its table (the edge kernel bounds a row's messages
        with it).  After rows of ``h`` were overwritten (halo refresh of the ghost rows) their entries are recomputed here
        -- device-side, no sync -- and the largest scaled value is returned (device scalar) for the caller's overflow guard."""
        premix = state[1]
        N = h.shape[0]
        ud = premix.numel() // N - 1
        hmax = premix[N * ud:]
        m = h[rows].float().abs().amax(1) if rows.numel() else h.new_zeros(0, dtype=torch.float32)
        if in_scale is not None and h.dtype == torch.float32:
            m = m * in_scale[0]
        hmax[rows] = m
        return m.max() if m.numel() else torch.zeros((), device=h.device)

    def executed_flops_per_edge(self) -> int:
        """Flops the edge kernel actually EXECUTES on the matrix pipe per edge and product pass: the dst half and the distance
        channel of product #1 are contracted once per NODE (pre-mix launch), so the edge kernel's K is H, not 2H + 1."""
        H, L = self.hidden, self.lmax
        fl = 0
        for _tp in (1, 2):
            for l3 in range(L + 1):
                M = H * (1 + L) if l3 == 0 else H
                for l1 in range(L + 1):
                    for l2 in range(L + 1):
                        if abs(l1 - l2) <= l3 <= l1 + l2 and (l1 + l2 + l3) % 2 == 0:
                            # mix-first paths: 2 l1 + 1 groups; feature-first (scalar outputs from l1 > 0): one group
                            groups = 1 if (l3 == 0 and l1 > 0) else (2 * l1 + 1)
                            fl += 2 * H * M * groups
        return fl

    def forward(self, h: torch.Tensor, g: RadiusGraph, msg1, msg2, in_scale: torch.Tensor | None = None, edges=None,
                cont=None, return_state: bool = False):
        """h [N, width] fp32 | bf16 (Morton order of ``g``) -> aggregated messages [N, width] in h's dtype (the sums are
        fp32 in both cases; bf16 storage rounds them once).

        ``edges = (src, dst)``: an explicit dst-sorted edge list instead of ``g``'s (sharding: interior / boundary edges).
        ``return_state``: return ``(out, state)`` with ``state = (out, premix)`` -- the pre-mix table (6.8 GB at 1 M
        particles) stays referenced only as long as the caller keeps it.
        ``cont``: such a state -- the call then ADDS its edges' messages to ``out`` (same ``h`` rows for every dst node
        required) and returns ``out``."""
        if not h.is_cuda or not self.supports(h.dtype):
            raise RuntimeError(f"fused message: ROCm tensor in float32 (or bfloat16 for hidden >= 32) required, got "
                               f"{h.dtype} on {h.device} (no CPU path)")
        io, code, esz = h.dtype, _lib.dtype_code(h.dtype), h.element_size()
        N, W = h.shape
        if W != self.width or N != g.rowptr.numel() - 1:
            raise RuntimeError(f"fused message: h must be [{g.rowptr.numel() - 1}, {self.width}], got {tuple(h.shape)}")
        if h.stride(-1) != 1 or (h.stride(0) * esz) % 16 or h.data_ptr() % 16:
            h = h.contiguous()
        dev = h.device
        lib = _lib.load()
        src, dst = (g.src, g.dst) if edges is None else edges
        E = int(src.numel())
        assert src.dtype == torch.int32 and dst.dtype == torch.int32 and dst.numel() == E
        if cont is None:
            out = torch.empty((N, W), dtype=torch.float32, device=dev)
        else:
            out = cont[0]
        if N == 0:
            return (out.to(io), None) if return_state else out.to(io)
        with torch.cuda.device(dev):
            packed = self.packed(msg1, msg2, dev, io)
            if in_scale is None and io == torch.float32:
                from . import ops
                in_scale = ops.pow2_scale([h])
            sc = in_scale.data_ptr() if (in_scale is not None and io == torch.float32) else None
            hd = self._plans.handle(dev)
            stream = torch.cuda.current_stream(dev).cuda_stream
            mode = "<bf16 storage, bf16 MFMA>" if io == torch.bfloat16 else "<fp16x3 split MFMA>"
            ud = int(lib.e3_msg_premix_floats_per_node(hd))
            if cont is not None:
                premix = cont[1]
            else:
                premix = torch.empty(N * ud, dtype=torch.float32, device=dev)
                t0 = profiling.begin() if profiling.enabled() else None
                _lib.check(lib.e3_msg_premix(hd, h.data_ptr(), h.stride(0), N, packed.data_ptr(), sc,
                                             premix.data_ptr(), code, stream), "e3_msg_premix")
                if t0 is not None:
                    # node-level GEMM h [N, (l_max+1)^2 H] x W_dst: reads h, writes the table
                    profiling.end(f"msg_premix lmax={self.lmax} H={self.hidden} N={N} {io}", N, N * (esz * W + 4 * ud), t0,
                                  flops=2 * self.hidden * ud * N, kernel="e3::msg_premix_kernel" + mode)
            t0 = profiling.begin() if profiling.enabled() else None
            _lib.check(lib.e3_msg_forward(hd, h.data_ptr(), h.stride(0), N, g.pos4.data_ptr(), src.data_ptr(),
                                          dst.data_ptr(), E, packed.data_ptr(), sc, premix.data_ptr(),
                                          out.data_ptr(), out.stride(0), code, 0 if cont is None else 1,
                                          int(self.tiles_per_block), stream), "e3_msg_forward")
            if t0 is not None:
                # algorithmic bytes: h read once, positions, the two index columns, aggregated rows written once
                nb = esz * N * W + 16 * N + 8 * E + 4 * N * W
                ws = bool(self.lmax == 2 and self.hidden == 32 and int(self.tiles_per_block) >= 0)
                profiling.end(f"msg_fused lmax={self.lmax} H={self.hidden} E={E} {io}", E, nb, t0,
                              flops=self.flops_per_edge() * E,
                              kernel=("e3::msg_ws_kernel" if ws else "e3::msg_fused_kernel") + mode,
                              executed_flops=self.executed_flops_per_edge() * E)
        # fp32 sums; callers in bf16 storage round once (SEGNNLayer)
        return (out, (out, premix)) if return_state else out
