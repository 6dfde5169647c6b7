"""CPU oracle for the L1 tensor-product hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain CPU restatement of the algorithm in
`/root/reference/models/segnn/l1_tensor_prod.py` (cited below as ``L1TP.py:<line>``).  It is the
checker for the HIP path: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it.  The product (``scalable-e3-gnn_amd/``) never does and
fails loudly when the HIP library is missing.

Parity pin: the restatement is checked (``tests/test_oracle_golden.py``) against
  * the known-answer vectors recorded in SURVEY.md §4 (norm tables, state-dict shapes, the
    RNG-free deterministic KAT), and
  * golden fixtures under ``tests/golden/*.npz`` produced by running the *unmodified* reference
    file in the build container (``tests/golden/make_golden.py``; e3nn is absent offline, so the
    run registers this repo's bookkeeping-only ``Irreps`` parser under the name ``e3nn.o3`` —
    e3nn contributes no arithmetic to the reference file, `L1TP.py:5`).
Parity with e3nn's own ``FullyConnectedTensorProduct`` is NOT pinned (e3nn unavailable).

Two forward restatements are provided:
  * ``forward_faithful``    – torch CPU, same op pattern as the reference (class gather → products
                              → concat → matmul → column scatter → norm).  Used as ``cpu_baseline``.
  * ``forward_closed_form`` – numpy, fp64 by default, explicit einsum of §8a-5..8 formulas.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

C3 = 1.0 / math.sqrt(3.0)  # cg110 = cg011, L1TP.py:92-93
C6 = 1.0 / math.sqrt(6.0)  # cg111, L1TP.py:94
SH_BLOCKS = [(0, 1, 1), (1, -1, 1)]  # Irreps.spherical_harmonics(1) = 1x0e+1x1o, L1TP.py:17

CLASSES = ("l0e", "l0o", "l1e", "l1o")


def _cls(l: int, p: int) -> str:
    return f"l{l}{'e' if p == 1 else 'o'}"


def parse_blocks(irreps) -> List[Tuple[int, int, int]]:
    """Accept 'AxLp+...' strings or any iterable of entries with .mul/.ir.l/.ir.p -> [(l,p,mul)]."""
    if isinstance(irreps, str):
        out = []
        for term in irreps.split("+"):
            term = term.strip()
            mul, ir = (term.split("x") if "x" in term else ("1", term))
            out.append((int(ir[:-1]), 1 if ir[-1] == "e" else -1, int(mul)))
        return out
    if irreps and isinstance(irreps[0], tuple) and len(irreps[0]) == 3:
        return [tuple(int(v) for v in b) for b in irreps]
    return [(int(m.ir.l), int(m.ir.p), int(m.mul)) for m in irreps]


def blocks_dim(blocks) -> int:
    return sum((2 * l + 1) * mul for l, p, mul in blocks)


# ------------------------------------------------------------------------------------------------
# a-1  column classification (L1TP.py:23-77)
# ------------------------------------------------------------------------------------------------
def class_columns(blocks) -> Dict[str, np.ndarray]:
    """Column indices of each (l,p) class, in order of appearance (the reference's bool masks,
    L1TP.py:24-36 / 53-65: an l=0 block marks ``mul`` columns, an l=1 block ``3*mul``)."""
    cols = {c: [] for c in CLASSES}
    i = 0
    for l, p, mul in blocks:
        width = (2 * l + 1) * mul
        if l in (0, 1):
            cols[_cls(l, p)].extend(range(i, i + width))
        i += width
    return {c: np.asarray(v, dtype=np.int64) for c, v in cols.items()}


@dataclass
class Layout:
    in1_blocks: List[Tuple[int, int, int]]
    out_blocks: List[Tuple[int, int, int]]
    in1_dim: int
    out_dim: int
    i1: Dict[str, np.ndarray]  # in1 columns per class
    o: Dict[str, np.ndarray]   # out columns per class
    n0e: int
    n0o: int
    n1e: int
    n1o: int
    # a-2: weight shapes (rows, cols) or None when the reference creates no parameter
    wshape: Dict[str, Optional[Tuple[int, int]]] = field(default_factory=dict)


def make_layout(in1_irreps, out_irreps=None) -> Layout:
    ib = parse_blocks(in1_irreps)
    ob = parse_blocks(out_irreps) if out_irreps is not None else list(ib)  # L1TP.py:18
    assert max(l for l, _, _ in ib) == 1  # L1TP.py:13
    if out_irreps is not None:
        assert max(l for l, _, _ in ob) == 1  # L1TP.py:14
    i1, o = class_columns(ib), class_columns(ob)
    n0e, n0o = len(i1["l0e"]), len(i1["l0o"])
    n1e, n1o = len(i1["l1e"]) // 3, len(i1["l1o"]) // 3
    rows = {  # L1TP.py:81-88, row order = forward concat order
        "l0e": n0e + n1o,
        "l0o": n0o + n1e,
        "l1e": n0o + n1e + n1o,
        "l1o": n0e + n1o + n1e,
    }
    cols = {"l0e": len(o["l0e"]), "l0o": len(o["l0o"]), "l1e": len(o["l1e"]) // 3, "l1o": len(o["l1o"]) // 3}
    outdim = {"l0e": len(o["l0e"]), "l0o": len(o["l0o"]), "l1e": len(o["l1e"]), "l1o": len(o["l1o"])}
    wshape = {c: ((rows[c], cols[c]) if rows[c] > 0 and outdim[c] > 0 else None) for c in CLASSES}
    return Layout(ib, ob, blocks_dim(ib), blocks_dim(ob), i1, o, n0e, n0o, n1e, n1o, wshape)


# ------------------------------------------------------------------------------------------------
# a-4  normalisation (L1TP.py:96-193) with quirks Q1..Q6 of SURVEY.md §8a-4
# ------------------------------------------------------------------------------------------------
@dataclass
class NormResult:
    is_norm: bool
    norms: Optional[Dict[str, np.ndarray]]        # fp32-rounded (Q4), per class, len = class dim
    a: Optional[List[float]]                      # unrounded per-out-entry factor (path_weight)
    wi: Optional[List[float]]                     # weight re-init bound per out entry
    instructions: Optional[List[tuple]]           # (i_in1, i_in2, i_out, 'uvw', True, a, (mul1, 1, mul_out))


def normalisation(in1_irreps, out_irreps=None, irrep_normalization="component",
                  path_normalization="element", in1_var=None, in2_var=None, out_var=None) -> NormResult:
    ib = parse_blocks(in1_irreps)
    ob = parse_blocks(out_irreps) if out_irreps is not None else list(ib)
    sb = SH_BLOCKS
    # L1TP.py:97-113
    in1_var = [1.0] * len(ib) if in1_var is None else [float(v) for v in in1_var]
    in2_var = [1.0] * len(sb) if in2_var is None else [float(v) for v in in2_var]
    out_var = [1.0] * len(ob) if out_var is None else [float(v) for v in out_var]
    assert len(in1_var) == len(ib), "Len of ir1_var must be equal to len(irreps_in1)"
    assert len(in2_var) == len(sb), "Len of ir2_var must be equal to len(irreps_in2)"
    assert len(out_var) == len(ob), "Len of out_var must be equal to len(irreps_out)"

    # L1TP.py:115-118
    is_norm = irrep_normalization in ("component", "norm") or path_normalization in ("element", "path")
    if not is_norm:
        return NormResult(False, None, None, None, None)  # Q2: forward then has no is_comp_norm
    is_comp = irrep_normalization != "norm" and path_normalization != "path"
    assert is_comp, "Not all norms are implemented yet."  # Q3

    alpha, x, ins = [], [], []
    for io, (lo, po, mo) in enumerate(ob):  # L1TP.py:122-151
        alpha.append((2 * lo + 1) * out_var[io] if irrep_normalization == "component" else 1)
        x.append(0.0 if path_normalization in ("element", "none") else 1)
        for i2, (l2, p2, m2) in enumerate(sb):
            for i1, (l1, p1, m1) in enumerate(ib):
                # Q1: python precedence makes this  A or (B and C)  — parity is only checked for l_out = 1
                if (lo == 0 and l2 == l1) or ((lo == 1 and (l2 | l1)) and (po == p2 * p1)):
                    if path_normalization in ("element", "none"):
                        x[-1] += in1_var[i1] * in2_var[i2] * m1 * m2
                    ins.append([i1, i2, io, "uvw", True, alpha[-1], (m1, m2, mo)])

    cls_dim = {c: 0 for c in CLASSES}
    for l, p, mul in ob:
        if l in (0, 1):
            cls_dim[_cls(l, p)] += (2 * l + 1) * mul
    norms = {c: np.empty(cls_dim[c], dtype=np.float32) for c in CLASSES}  # Q4: fp32 buffers (L1TP.py:159-162)
    pos = {c: 0 for c in CLASSES}
    a_list, wi_list = [], []
    for io, ((lo, po, mo), ai, xi) in enumerate(zip(ob, alpha, x)):  # L1TP.py:164-193
        if path_normalization == "none":
            a = math.sqrt(ai)
            wi = 1 / math.sqrt(xi)  # Q6: ZeroDivisionError when xi == 0
        else:
            a = math.sqrt((ai / xi) if xi > 0 else ai)
            wi = 1
        a_list.append(a)
        wi_list.append(wi)
        c = _cls(lo, po)
        width = (2 * lo + 1) * mo
        norms[c][pos[c]:pos[c] + width] = a
        pos[c] += width
        for inst in ins:
            if inst[2] == io:
                inst[5] = a
    return NormResult(True, norms, a_list, wi_list, [tuple(i) for i in ins])


def weight_reinit_slices(out_irreps_blocks) -> List[Tuple[str, int, int, int]]:
    """Column slices the reference re-draws with ``uniform_(-wi, wi)`` (L1TP.py:171-189), in call
    order: (class, col_start, col_stop, out_entry_index).  Q5: for l=1 classes the running index
    advances by ``3*mul`` but slices weight *columns* ``[i, i+mul)``; python slicing clamps."""
    pos = {c: 0 for c in CLASSES}
    out = []
    for io, (lo, po, mo) in enumerate(out_irreps_blocks):
        c = _cls(lo, po)
        out.append((c, pos[c], pos[c] + mo, io))
        pos[c] += (2 * lo + 1) * mo
    return out


# ------------------------------------------------------------------------------------------------
# a-5..a-9  forward
# ------------------------------------------------------------------------------------------------
def forward_closed_form(lay: Layout, in1, in2, W: Dict[str, np.ndarray],
                        norms: Optional[Dict[str, np.ndarray]], dtype=np.float64) -> np.ndarray:
    """numpy restatement of L1TP.py:240-299 (einsum form).  ``in2`` may be [B,4] or [1,4]."""
    in1 = np.asarray(in1, dtype=dtype)
    in2 = np.asarray(in2, dtype=dtype)
    B = in1.shape[0]
    assert in1.shape[-1] == lay.in1_dim and in2.shape[-1] == 4
    y0 = in2[:, 0:1]                       # [B|1,1]
    y1 = in2[:, 1:4]                       # [B|1,3]
    s0e = in1[:, lay.i1["l0e"]]
    s0o = in1[:, lay.i1["l0o"]]
    v1e = in1[:, lay.i1["l1e"]].reshape(B, lay.n1e, 3)
    v1o = in1[:, lay.i1["l1o"]].reshape(B, lay.n1o, 3)
    out = np.zeros((B, lay.out_dim), dtype=dtype)

    y1full = np.broadcast_to(y1, (B, 3))
    y0full = np.broadcast_to(y0, (B, 1))

    def nrm(c):
        return 1.0 if norms is None else np.asarray(norms[c], dtype=dtype)

    if len(lay.o["l0e"]) > 0:  # L1TP.py:242-256
        f = [s0e * y0]
        if lay.n1o > 0:
            f.append(C3 * np.einsum("bkc,bc->bk", v1o, y1full))
        out[:, lay.o["l0e"]] = (np.concatenate(f, -1) @ np.asarray(W["l0e"], dtype)) * nrm("l0e")
    if len(lay.o["l0o"]) > 0:  # L1TP.py:258-269
        f = [s0o * y0]
        if lay.n1e > 0:
            f.append(C3 * np.einsum("bkc,bc->bk", v1e, y1full))
        out[:, lay.o["l0o"]] = (np.concatenate(f, -1) @ np.asarray(W["l0o"], dtype)) * nrm("l0o")
    if len(lay.o["l1e"]) > 0:  # L1TP.py:271-284
        f = [C3 * s0o[:, :, None] * y1full[:, None, :]]
        if lay.n1e > 0:
            f.append(C3 * v1e * y0full[:, :, None])
        if lay.n1o > 0:
            f.append(C6 * np.cross(v1o, y1full[:, None, :]))
        F = np.concatenate(f, 1)  # [B,K,3]
        o = np.einsum("bkc,kw->bwc", F, np.asarray(W["l1e"], dtype)).reshape(B, len(lay.o["l1e"]))
        out[:, lay.o["l1e"]] = o * nrm("l1e")
    if len(lay.o["l1o"]) > 0:  # L1TP.py:286-297
        f = [C3 * s0e[:, :, None] * y1full[:, None, :]]
        if lay.n1o > 0:
            f.append(C3 * v1o * y0full[:, :, None])
        if lay.n1e > 0:
            f.append(C6 * np.cross(v1e, y1full[:, None, :]))
        F = np.concatenate(f, 1)
        o = np.einsum("bkc,kw->bwc", F, np.asarray(W["l1o"], dtype)).reshape(B, len(lay.o["l1o"]))
        out[:, lay.o["l1o"]] = o * nrm("l1o")
    return out


def forward_faithful(lay: Layout, in1, in2, W, norms):
    """torch-CPU restatement with the reference's op pattern (L1TP.py:234-299): boolean-mask style
    class gathers, broadcast products, ``cat``, ``@`` / ``tensordot``, masked column assignment,
    masked in-place norm multiply.  Arithmetic order per element matches the reference, so fp32
    results agree to rounding of the BLAS reduction order.  This is the ``cpu_baseline`` kernel."""
    import torch

    assert in1.shape[-1] == lay.in1_dim, f"Incorrect last dimension for in1 = {in1.shape[-1]}, required is {lay.in1_dim}"
    assert in2.shape[-1] == 4, f"Incorrect last dimension for in2 = {in2.shape[-1]}, required is 4"
    idx = {k: torch.as_tensor(v) for k, v in lay.i1.items()}
    odx = {k: torch.as_tensor(v) for k, v in lay.o.items()}
    sh0 = torch.tensor([0])
    sh1 = torch.tensor([1, 2, 3])
    out = torch.empty((in1.shape[0], lay.out_dim), dtype=in1.dtype)
    has_norm = norms is not None

    def scalar_block(s_cls, v_cls, nvec):
        parts = [in1[:, idx[s_cls]] * in2[:, sh0]]
        if nvec > 0:
            parts.append(C3 * torch.linalg.vecdot(in1[:, idx[v_cls]].reshape((-1, nvec, 3)),
                                                  in2[:, None, sh1]).reshape(-1, nvec))
        return torch.cat(parts, -1)

    def vector_block(s_cls, v_same, n_same, v_other, n_other):
        parts = [C3 * in1[:, idx[s_cls], None] * in2[:, None, sh1]]
        if n_same > 0:
            parts.append(C3 * in1[:, idx[v_same]].reshape(-1, n_same, 3) * in2[:, None, sh0])
        if n_other > 0:
            parts.append(C6 * torch.linalg.cross(in1[:, idx[v_other]].reshape(-1, n_other, 3), in2[:, None, sh1]))
        return torch.cat(parts, -2)

    if len(odx["l0e"]) > 0:
        out[:, odx["l0e"]] = (scalar_block("l0e", "l1o", lay.n1o) @ W["l0e"]).to(dtype=out.dtype)
        if has_norm:
            out[:, odx["l0e"]] *= norms["l0e"]
    if len(odx["l0o"]) > 0:
        out[:, odx["l0o"]] = (scalar_block("l0o", "l1e", lay.n1e) @ W["l0o"]).to(dtype=out.dtype)
        if has_norm:
            out[:, odx["l0o"]] *= norms["l0o"]
    if len(odx["l1e"]) > 0:
        F = vector_block("l0o", "l1e", lay.n1e, "l1o", lay.n1o)
        out[:, odx["l1e"]] = torch.tensordot(F, W["l1e"], ([-2], [0])).transpose(-1, -2).reshape(-1, len(odx["l1e"]))
        if has_norm:
            out[:, odx["l1e"]] *= norms["l1e"]
    if len(odx["l1o"]) > 0:
        F = vector_block("l0e", "l1o", lay.n1o, "l1e", lay.n1e)
        out[:, odx["l1o"]] = torch.tensordot(F, W["l1o"], ([-2], [0])).transpose(-1, -2).reshape(-1, len(odx["l1o"]))
        if has_norm:
            out[:, odx["l1o"]] *= norms["l1o"]
    return out.contiguous()
