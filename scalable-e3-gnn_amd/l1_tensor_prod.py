"""``L1TensorProduct`` — drop-in host mirror of the reference operator, running on the HIP library.

Mirrors `/root/reference/models/segnn/l1_tensor_prod.py` (``L1TP.py``): constructor signature
(`L1TP.py:9-11`), public attributes (`:16-21,24-77,91-94,115-117,121`), parameter / buffer names and
shapes (`:81-88,159-162`), initial-weight RNG stream (`:82-88,175-188`), normalisation including its
quirks Q1–Q6, the error behaviour of ctor and forward (`:13-14,101-118,236-237`)
and ``forward(in1, in2) -> Tensor`` (`:234-299`).

What differs by design: ``forward`` does not run a chain of ATen gathers/cats/matmuls — it makes one
call into ``libe3gnn_hip.so`` (fused gfx950 kernel) through the C ABI in ``include/e3gnn.h``.  There
is no CPU implementation here: CPU tensors raise, and so does a missing library.
"""
from __future__ import annotations

import ctypes
from math import sqrt
from typing import List, Optional

import torch
from torch import Tensor
from torch.nn import Module, Parameter

from . import _lib, profiling
from .irreps import Instruction, Irreps, as_blocks

_CLS = ("l0e", "l0o", "l1e", "l1o")


def _cls_name(l: int, p: int) -> str:
    return f"l{l}{'e' if p == 1 else 'o'}"


def _class_masks(irreps) -> dict:
    """Boolean column masks per (l,p) class, in declaration order (`L1TP.py:24-36,53-65`)."""
    blocks = as_blocks(irreps)
    dim = sum((2 * l + 1) * m for l, _, m in blocks)
    masks = {c: torch.zeros(dim, dtype=torch.bool) for c in _CLS}
    col = 0
    for l, p, mul in blocks:
        width = (2 * l + 1) * mul
        if l <= 1:
            masks[_cls_name(l, p)][col:col + width] = True
        col += width
    return masks


def _Plan(in1_blocks, out_blocks) -> "_lib.DevicePlans":
    """Per-device ``e3_l1tp_plan*`` handles (deep-copy / pickle safe)."""
    return _lib.DevicePlans("e3_l1tp_plan_create", "e3_l1tp_plan_destroy", [tuple(b) for b in in1_blocks],
                            [tuple(b) for b in out_blocks])


class _L1TPFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod: "L1TensorProduct", in1: Tensor, in2: Tensor, *weights: Optional[Tensor]):
        out = mod._hip_forward(in1, in2)
        ctx.mod = mod
        # the weights travel through save_for_backward as well: autograd's version counters then catch an in-place
        # update between forward and backward, and backward differentiates the values the forward used
        ctx.wpresent = [w is not None for w in weights]
        ctx.save_for_backward(in1, in2, *[w for w in weights if w is not None])
        return out

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        mod = ctx.mod
        in1, in2, *saved_w = ctx.saved_tensors
        it = iter(saved_w)
        ws = [next(it) if present else None for present in ctx.wpresent]
        need_in1, need_in2 = ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        need_w = any(ctx.needs_input_grad[3:])
        g_in1, g_in2, g_w = mod._hip_backward(in1, in2, grad_out.contiguous(), need_in1, need_in2, need_w, weights=ws)
        gw_out = []
        for i, c in enumerate(_CLS):
            gw_out.append(g_w[c] if (ctx.needs_input_grad[3 + i] and g_w is not None) else None)
        return (None, g_in1, g_in2, *gw_out)


class L1TensorProduct(Module):
    """``feature (x) Y_{l<=1}(edge) -> feature`` with fully-connected learned weights, l <= 1.

    Same constructor and semantics as the reference class (`L1TP.py:8-299`).  ``in1_irreps`` /
    ``out_irreps`` may be ``e3nn.o3.Irreps``, this package's ``Irreps`` or (extension) a string.
    """

    def __init__(self, in1_irreps, out_irreps=None,
                 irrep_normalization="component", path_normalization="element",
                 in1_var: List[float] = None, in2_var: List[float] = None, out_var: List[float] = None) -> None:
        super().__init__()
        if isinstance(in1_irreps, str):
            in1_irreps = Irreps(in1_irreps)
        if isinstance(out_irreps, str):
            out_irreps = Irreps(out_irreps)
        assert in1_irreps.lmax == 1                                   # L1TP.py:13 (bare assert)
        if out_irreps is not None:
            assert out_irreps.lmax == 1                               # L1TP.py:14

        self.iri1 = in1_irreps
        self.iri2 = Irreps.spherical_harmonics(1)                     # 1x0e+1x1o, L1TP.py:17
        self.iro = out_irreps if out_irreps is not None else in1_irreps
        self.in1_dim = self.iri1.dim
        self.in2_dim = self.iri2.dim

        # column masks: plain CPU tensors, deliberately not buffers (as in the reference)
        m1, m2, mo = _class_masks(self.iri1), _class_masks(self.iri2), _class_masks(self.iro)
        self.iri1_l0e, self.iri1_l0o, self.iri1_l1e, self.iri1_l1o = (m1[c] for c in _CLS)
        self.iri2_l0e, self.iri2_l1o = m2["l0e"], m2["l1o"]
        self.iro_l0e, self.iro_l0o, self.iro_l1e, self.iro_l1o = (mo[c] for c in _CLS)

        # counts as python ints (L1TP.py:67-77)
        self.num_i1_l0e = int(m1["l0e"].sum())
        self.num_i1_l0o = int(m1["l0o"].sum())
        self.num_i1_l0 = self.num_i1_l0e + self.num_i1_l0o
        self.dim_i1_l1e = int(m1["l1e"].sum())
        self.num_i1_l1e = self.dim_i1_l1e // 3
        self.dim_i1_l1o = int(m1["l1o"].sum())
        self.num_i1_l1o = self.dim_i1_l1o // 3
        self.dim_o_l0e = int(mo["l0e"].sum())
        self.dim_o_l0o = int(mo["l0o"].sum())
        self.dim_o_l1e = int(mo["l1e"].sum())
        self.dim_o_l1o = int(mo["l1o"].sum())

        # parameters: rows follow the concat order of the forward, U[-1,1] first (L1TP.py:81-88).
        # The RNG calls are issued in the reference's order so equal seeds give equal weights.
        n0e, n0o, n1e, n1o = self.num_i1_l0e, self.num_i1_l0o, self.num_i1_l1e, self.num_i1_l1o
        shapes = {
            "l0e": (n0e + n1o, self.dim_o_l0e, self.dim_o_l0e),
            "l0o": (n0o + n1e, self.dim_o_l0o, self.dim_o_l0o),
            "l1e": (n0o + n1e + n1o, self.dim_o_l1e // 3, self.dim_o_l1e),
            "l1o": (n0e + n1o + n1e, self.dim_o_l1o // 3, self.dim_o_l1o),
        }
        for c in _CLS:
            rows, cols, odim = shapes[c]
            if rows > 0 and odim > 0:
                setattr(self, "weights_" + c, Parameter(torch.rand((rows, cols)) * 2 - 1))

        self.cg000 = 1
        self.cg110 = 1 / sqrt(3)
        self.cg011 = self.cg110
        self.cg111 = 1 / sqrt(6)

        def _vars(given, n, msg):
            if given is None:
                return [1.0] * n
            given = [float(v) for v in given]
            assert len(given) == n, msg
            return given

        in1_var = _vars(in1_var, len(self.iri1), "Len of ir1_var must be equal to len(irreps_in1)")
        in2_var = _vars(in2_var, len(self.iri2), "Len of ir2_var must be equal to len(irreps_in2)")
        out_var = _vars(out_var, len(self.iro), "Len of out_var must be equal to len(irreps_out)")

        self._plan = None
        self._tpplan = None
        self._packed = None
        self._packed_key = None
        self.kernel = 0  # 0 auto, 1 generic, 2 MFMA (see e3_l1tp_forward)

        self.is_norm = irrep_normalization in ("component", "norm") or path_normalization in ("element", "path")
        if not self.is_norm:
            return  # Q2: forward then fails on the missing `is_comp_norm`, as the reference does
        self.is_comp_norm = irrep_normalization != "norm" and path_normalization != "path"
        torch._assert(self.is_comp_norm, "Not all norms are implemented yet.")   # Q3

        self._init_normalisation(irrep_normalization, path_normalization, in1_var, in2_var, out_var)

    # ------------------------------------------------------------------------------------------
    def _init_normalisation(self, irrep_normalization, path_normalization, in1_var, in2_var, out_var):
        """Norm buffers, weight re-draw and ``instructions`` (`L1TP.py:120-193`).

        Path counting reproduces quirk Q1: because of operator precedence in the reference
        predicate (`L1TP.py:137-138`) a scalar output counts every (in1, in2) pair of equal ``l``
        regardless of parity, while vector outputs are parity-checked.
        """
        iri1, iri2, iro = as_blocks(self.iri1), as_blocks(self.iri2), as_blocks(self.iro)
        counted = path_normalization in ("element", "none")
        self.instructions: List[Instruction] = []
        alpha, x = [], []
        for io, (lo, po, mo) in enumerate(iro):
            alpha.append((2 * lo + 1) * out_var[io] if irrep_normalization == "component" else 1)
            x.append(0.0 if counted else 1)
            for i2, (l2, p2, m2) in enumerate(iri2):
                for i1, (l1, p1, m1) in enumerate(iri1):
                    scalar_path = lo == 0 and l2 == l1                       # parity-blind (Q1)
                    vector_path = lo == 1 and bool(l2 | l1) and po == p2 * p1
                    if scalar_path or vector_path:
                        if counted:
                            x[-1] += in1_var[i1] * in2_var[i2] * m1 * m2
                        self.instructions.append(Instruction(i1, i2, io, "uvw", True, alpha[-1], (m1, m2, mo)))

        # Q4: buffers are created fp32 and filled from python doubles
        for c, d in zip(_CLS, (self.dim_o_l0e, self.dim_o_l0o, self.dim_o_l1e, self.dim_o_l1o)):
            self.register_buffer("norm_" + c, torch.empty(d))
        cursor = {c: 0 for c in _CLS}
        for io, ((lo, po, mo), ai, xi) in enumerate(zip(iro, alpha, x)):
            if path_normalization == "none":
                a, wi = sqrt(ai), 1 / sqrt(xi)                               # Q6: ZeroDivisionError if xi == 0
            else:
                a, wi = sqrt((ai / xi) if xi > 0 else ai), 1
            c = _cls_name(lo, po)
            i, width = cursor[c], (2 * lo + 1) * mo
            with torch.no_grad():
                getattr(self, "norm_" + c)[i:i + width] = a
                # Q5: weight *columns* are sliced [i, i+mul) although i advances by `width`;
                # a missing parameter raises AttributeError here exactly as in the reference.
                getattr(self, "weights_" + c)[:, i:i + mo].uniform_(-wi, wi)
            cursor[c] = i + width
            self.instructions = [
                ins._replace(path_weight=a) if ins.i_out == io else ins for ins in self.instructions
            ]

    # ------------------------------------------------------------------------------------------
    # HIP path
    # ------------------------------------------------------------------------------------------
    def _weights(self) -> List[Optional[Tensor]]:
        return [getattr(self, "weights_" + c, None) for c in _CLS]

    def _norms(self) -> List[Optional[Tensor]]:
        return [getattr(self, "norm_" + c, None) for c in _CLS]

    def _get_plan(self):
        if self._plan is None:
            self._plan = _Plan(as_blocks(self.iri1), as_blocks(self.iro))
        return self._plan

    def _packed_weights(self, dtype, device) -> Tensor:
        """Packed weight buffer for the kernels; rebuilt when any parameter/buffer changed."""
        ws, ns = self._weights(), self._norms()
        key = (dtype, device) + tuple((t.data_ptr(), t._version) if t is not None else None for t in ws + ns)
        stream = torch.cuda.current_stream(device)
        if self._packed is not None and self._packed_key == key:
            # packed on another stream: this stream waits for that pack launch (same rule as message.py / tensor_product.py)
            if self._packed_stream != stream.cuda_stream:
                stream.wait_event(self._packed_event)
            return self._packed
        lib = _lib.load()
        code = _lib.dtype_code(dtype)
        for t in ws + ns:
            if t is not None and t.numel() > 0 and (t.dtype != dtype or t.device != device):
                raise RuntimeError(
                    f"L1TensorProduct: parameter/buffer dtype/device {t.dtype}/{t.device} does not match "
                    f"input {dtype}/{device} (cast the module, autocast is not supported — as in the reference)")
        plan = self._get_plan()
        nbytes = lib.e3_l1tp_packed_bytes(plan.handle(device), code)
        packed = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)
        wsc = [w.detach().contiguous() if w is not None else None for w in ws]
        nsc = [n.detach().contiguous() if n is not None else None for n in ns]
        _lib.check(lib.e3_l1tp_pack_weights(plan.handle(device), _lib.ptr4(wsc), _lib.ptr4(nsc), code,
                                            packed.data_ptr(), stream.cuda_stream), "e3_l1tp_pack_weights")
        ev = torch.cuda.Event()
        ev.record(stream)
        self._packed, self._packed_key, self._packed_stream, self._packed_event = packed, key, stream.cuda_stream, ev
        return packed

    def _hip_forward(self, in1: Tensor, in2: Tensor) -> Tensor:
        lib = _lib.load()
        B = in1.shape[0]
        out = torch.empty((B, len(self.iro_l0e)), device=in1.device, dtype=in1.dtype)   # L1TP.py:240
        if B == 0:
            return out
        if in1.stride(-1) != 1:
            in1 = in1.contiguous()
        if in2.stride(-1) != 1:
            in2 = in2.contiguous()
        ld2 = 0 if (in2.shape[0] == 1 and B != 1) else in2.stride(0)
        with torch.cuda.device(in1.device):
            packed = self._packed_weights(in1.dtype, in1.device)
            stream = torch.cuda.current_stream(in1.device).cuda_stream
            t0 = profiling.begin() if profiling.enabled() else None
            _lib.check(lib.e3_l1tp_forward(self._get_plan().handle(in1.device), in1.data_ptr(), in1.stride(0),
                                           in2.data_ptr(), ld2, packed.data_ptr(), out.data_ptr(), out.stride(0),
                                           B, _lib.dtype_code(in1.dtype), int(self.kernel), stream),
                       "e3_l1tp_forward")
            if t0 is not None:
                tag = f"l1tp_fwd {self.iri1}->{self.iro} B={B}"
                n0e, n0o, n1e, n1o = self.num_i1_l0e, self.num_i1_l0o, self.num_i1_l1e, self.num_i1_l1o
                fl = (2 * (n0e + n1o) * self.dim_o_l0e + 2 * (n0o + n1e) * self.dim_o_l0o +
                      2 * (n0o + 3 * n1e + 3 * n1o) * (self.dim_o_l1e // 3) + 2 * (n0e + 3 * n1o + 3 * n1e) * (self.dim_o_l1o // 3))
                profiling.end(tag, B, in1.element_size() * (self.in1_dim + 4 + out.shape[1]) * B, t0, flops=fl * B,
                              kernel="e3::l1tp_fwd_mfma_kernel")
        return out

    def _hip_backward(self, in1, in2, grad_out, need_in1, need_in2, need_w, weights=None):
        lib = _lib.load()
        B = in1.shape[0]
        dtype, device = in1.dtype, in1.device
        ws, ns = (weights if weights is not None else self._weights()), self._norms()
        g_in1 = torch.empty_like(in1, memory_format=torch.contiguous_format) if need_in1 else None
        g_in2 = torch.empty_like(in2, memory_format=torch.contiguous_format) if need_in2 else None
        g_w = {c: (torch.empty_like(w) if (need_w and w is not None) else None) for c, w in zip(_CLS, ws)}
        if B == 0:
            for t in [g_in1, g_in2] + list(g_w.values()):
                if t is not None:
                    t.zero_()
            return g_in1, g_in2, g_w
        if in1.stride(-1) != 1:
            in1 = in1.contiguous()
        if in2.stride(-1) != 1:
            in2 = in2.contiguous()
        ld2 = 0 if (in2.shape[0] == 1 and B != 1) else in2.stride(0)
        code = _lib.dtype_code(dtype)
        from . import tensor_product as _tp
        if (B >= _tp._BWD_GEMM_MIN_ROWS and dtype == torch.float32 and need_in1 and not need_in2 and ld2 != 0 and
                getattr(self, "is_comp_norm", False)):
            # grad_in1 alone (the training case: harmonics of fixed positions) = THIS operator on the transposed plan:
            # C(l1,l2,l3)[a,b,c] = +-C(l3,l2,l1)[c,b,a], so grad_in1 = dual(grad_out * norm, in2) with the weight blocks
            # transposed (and the cross-product block negated) -- one launch of the MFMA forward kernel, nothing of size
            # [B, D3, K] in HBM.  grad_W still comes from the operand pass + one batched GEMM per class.
            g1 = self._dual_forward(ws, ns, in1, in2, grad_out)
            gws = [None] * 6
            if need_w:
                tpp = self._fused_plan()
                ws6, ns6 = list(ws) + [None, None], list(ns) + [None, None]
                with torch.cuda.device(device):
                    packed = tpp.packed(ws6, ns6, dtype, device)
                _, _, gws = _tp.tp_backward(tpp, packed, in1, in2, grad_out, ws6, False, False,
                                            [w is not None for w in ws6])
            return (g1, None, {c: (gws[i].to(ws[i].dtype) if gws[i] is not None else None) for i, c in enumerate(_CLS)})
        if B >= _tp._BWD_GEMM_MIN_ROWS and dtype in (torch.float32, torch.float64):
            # large B: operands -> library GEMMs -> contract on the general plan (this operator is its lmax_sh = 1 case:
            # same class order, weight-row order and norms, see forward_fused)
            tpp = self._fused_plan()
            ws6, ns6 = list(ws) + [None, None], list(ns) + [None, None]
            with torch.cuda.device(device):
                packed = tpp.packed(ws6, ns6, dtype, device)
            g1, g2, gws = _tp.tp_backward(tpp, packed, in1, in2, grad_out, ws6, need_in1, need_in2,
                                          [need_w and w is not None for w in ws6])
            return (g1, g2.to(in2.dtype) if g2 is not None else None,
                    {c: (gws[i].to(ws[i].dtype) if gws[i] is not None else None) for i, c in enumerate(_CLS)})
        with torch.cuda.device(device):
            plan = self._get_plan()
            wbytes = lib.e3_l1tp_backward_workspace_bytes(plan.handle(device), B, code)
            work = torch.empty(max(int(wbytes), 16), dtype=torch.uint8, device=device)
            wsc = [w.detach().contiguous() if w is not None else None for w in ws]
            nsc = [n.detach().contiguous() if n is not None else None for n in ns]
            stream = torch.cuda.current_stream(device).cuda_stream
            _lib.check(lib.e3_l1tp_backward(
                plan.handle(device), in1.data_ptr(), in1.stride(0), in2.data_ptr(), ld2,
                _lib.ptr4(wsc), _lib.ptr4(nsc), grad_out.data_ptr(), grad_out.stride(0),
                g_in1.data_ptr() if g_in1 is not None else None, g_in1.stride(0) if g_in1 is not None else 0,
                g_in2.data_ptr() if g_in2 is not None else None,
                _lib.ptr4([g_w[c] for c in _CLS]), work.data_ptr(), B, code, stream), "e3_l1tp_backward")
        return g_in1, g_in2, g_w

    def _dual_forward(self, ws, ns, in1, in2, grad_out):
        """grad_in1 of the forward as a forward of the dual operator (irreps swapped, lmax 1).  Weight rows of the forward
        (`L1TP.py:81-88`): l0e [0e | 1o], l0o [0o | 1e], l1e [0o | 1e | 1o x], l1o [0e | 1o | 1e x]; the dual's rows are the
        same lists with the multiplicities of the OUT irreps, each block the transpose of the forward block that connects the
        two classes, the cross-product blocks (`cg111`, antisymmetric) negated.  Norms leave with grad_out."""
        dual = getattr(self, "_dual", None)
        dev = in1.device
        if dual is None:
            with torch.random.fork_rng(devices=[]):   # the constructor draws weights: leave the caller's RNG stream alone
                dual = L1TensorProduct(self.iro, self.iri1)
            for c in _CLS:
                getattr(dual, "norm_" + c).fill_(1.0)
            dual.requires_grad_(False)
            object.__setattr__(self, "_dual", dual)   # not a submodule: no parameters of its own, rebuilt from self's
        if dual._weights()[0] is None and dual._weights()[1] is None and dual._weights()[2] is None and dual._weights()[3] is None:
            return torch.zeros_like(in1, memory_format=torch.contiguous_format)
        if next((w for w in dual._weights() if w is not None)).device != dev or \
                next((w for w in dual._weights() if w is not None)).dtype != in1.dtype:
            dual.to(device=dev, dtype=in1.dtype)
        n0e, n0o, n1e, n1o = self.num_i1_l0e, self.num_i1_l0o, self.num_i1_l1e, self.num_i1_l1o
        W = dict(zip(_CLS, ws))

        def blk(c, r0, n):   # rows r0 .. r0 + n of the forward matrix of class c, transposed; zeros when the class is absent
            w = W[c]
            return None if (w is None or n == 0) else w.detach()[r0:r0 + n].t()

        rows = {   # dual class -> [(block or None, rows of the block = multiplicity of the dual's in class), ...]
            "l0e": [(blk("l0e", 0, n0e), dual.num_i1_l0e, 1.0), (blk("l1o", 0, n0e), dual.num_i1_l1o, 1.0)],
            "l0o": [(blk("l0o", 0, n0o), dual.num_i1_l0o, 1.0), (blk("l1e", 0, n0o), dual.num_i1_l1e, 1.0)],
            "l1e": [(blk("l0o", n0o, n1e), dual.num_i1_l0o, 1.0), (blk("l1e", n0o, n1e), dual.num_i1_l1e, 1.0),
                    (blk("l1o", n0e + n1o, n1e), dual.num_i1_l1o, -1.0)],
            "l1o": [(blk("l0e", n0e, n1o), dual.num_i1_l0e, 1.0), (blk("l1o", n0e, n1o), dual.num_i1_l1o, 1.0),
                    (blk("l1e", n0o + n1e, n1o), dual.num_i1_l1e, -1.0)],
        }
        with torch.no_grad():
            for c in _CLS:
                wd = getattr(dual, "weights_" + c, None)
                if wd is None:
                    continue
                r = 0
                for b, nrows, sign in rows[c]:
                    if nrows == 0:
                        continue
                    if b is None:
                        wd[r:r + nrows].zero_()
                    else:
                        wd[r:r + nrows].copy_(b if sign > 0 else -b)
                    r += nrows
                assert r == wd.shape[0], (c, r, tuple(wd.shape))
            # grad_out * norm, per output column (class norms scattered to the out layout)
            key = tuple((t.data_ptr(), t._version) if t is not None else None for t in ns) + (dev, in1.dtype)
            if getattr(self, "_normcol_key", None) != key:
                full = torch.ones(len(self.iro_l0e), dtype=in1.dtype, device=dev)
                for c, n in zip(_CLS, ns):
                    m = getattr(self, "iro_" + c)
                    if n is not None and n.numel() and bool(m.any()):
                        full[m.to(dev)] = n.to(in1.dtype)
                object.__setattr__(self, "_normcol", full)
                object.__setattr__(self, "_normcol_key", key)
            gt = grad_out * self._normcol
            return dual._hip_forward(gt, in2)

    # ------------------------------------------------------------------------------------------
    # Fused message-function form (builder-defined extension; same arithmetic, same weights): the row gather
    # / concat of in1 and the SEGNN gate run inside the kernel (e3_tp_forward_fused, lmax_sh = 1).
    def _fused_plan(self):
        if getattr(self, "_tpplan", None) is None:
            from .tensor_product import TPPlan
            self._tpplan = TPPlan(self.iri1, self.iro, 1)
        return self._tpplan

    def fused_supported(self, gate: bool) -> bool:
        return self._fused_plan().fused_supported(gate)

    def forward_fused(self, segments, in2: Tensor, gate: bool = False, scatter=None, in_scale=None, residual=None,
                      out_scale=None):
        """``scatter=(row_node, n_nodes)``: fused segment-sum (see ``TPPlan.forward_fused``); returns None when the
        library has no such kernel for this product (callers then run the two kernels).  ``residual`` / ``out_scale``:
        residual add and operand scale of the result in the kernel's epilogue (-> (out, scale) with ``out_scale``)."""
        ws = self._weights() + [None, None]
        ns = self._norms() + [None, None]
        return self._fused_plan().forward_fused(ws, ns, segments, in2, gate, tag=f"{self.iri1}->{self.iro}",
                                                scatter=scatter, in_scale=in_scale, residual=residual, out_scale=out_scale)

    # ------------------------------------------------------------------------------------------
    def forward(self, in1: Tensor, in2: Tensor) -> Tensor:
        torch._assert(in1.shape[-1] == self.in1_dim,
                      f"Incorrect last dimension for in1 = {in1.shape[-1]}, required is {self.in1_dim}")
        torch._assert(in2.shape[-1] == self.in2_dim,
                      f"Incorrect last dimension for in2 = {in2.shape[-1]}, required is {self.in2_dim}")
        if in1.dim() != 2:
            # the reference's boolean-mask gather raises IndexError for anything but [B, D]
            raise IndexError(f"L1TensorProduct expects in1 of shape [B, {self.in1_dim}], got {tuple(in1.shape)}")
        if in2.dim() != 2 or in2.shape[0] not in (1, in1.shape[0]):
            raise RuntimeError(f"in2 must be [B, 4] or [1, 4], got {tuple(in2.shape)} for B = {in1.shape[0]}")
        self.is_comp_norm  # Q2: AttributeError when built with both normalisations "none"
        if not in1.is_cuda:
            raise RuntimeError(
                "L1TensorProduct (MI355X build) runs on ROCm tensors only; there is no CPU path. "
                "Move the module and its inputs to the GPU.")
        if in2.dtype != in1.dtype or in2.device != in1.device:
            raise RuntimeError(f"in1/in2 dtype or device mismatch: {in1.dtype}/{in1.device} vs {in2.dtype}/{in2.device}")
        return _L1TPFunction.apply(self, in1, in2, *self._weights())
