"""numpy fp64 oracle of the general SH tensor product (l <= 2).  TEST INFRASTRUCTURE ONLY.
Builder-defined contract (include/e3gnn.h "General SH tensor product"); reduces to the reference-pinned
``l1tp_oracle`` for l <= 1 (checked in tests/test_tp_l2.py).  Parity w.r.t. upstream/e3nn: unpinned."""
import numpy as np

from . import cg as CG
from .l1tp_oracle import parse_blocks

CLASSES = ("l0e", "l0o", "l1e", "l1o", "l2e", "l2o")


def cls_of(l, p):
    return 2 * l + (0 if p == 1 else 1)


def class_columns(blocks):
    """per class: array [channels, 2l+1] of column indices, channels in order of appearance"""
    cols = {c: [] for c in range(6)}
    i = 0
    for l, p, mul in blocks:
        for k in range(mul):
            cols[cls_of(l, p)].append(list(range(i + k * (2 * l + 1), i + (k + 1) * (2 * l + 1))))
        i += (2 * l + 1) * mul
    return {c: np.asarray(v, dtype=np.int64).reshape(len(v), 2 * (c >> 1) + 1) for c, v in cols.items()}


def paths(c3, n_in, lmax_sh):
    l3, p3 = c3 >> 1, (1 if c3 % 2 == 0 else -1)
    out = []
    for l1 in range(3):
        for l2 in range(lmax_sh + 1):
            if abs(l1 - l2) <= l3 <= l1 + l2:
                c1 = cls_of(l1, p3 * (-1) ** l2)
                if n_in[c1] > 0:
                    out.append((c1, l1, l2))
    return out


def shapes(in1_irreps, out_irreps, lmax_sh):
    ib, ob = parse_blocks(in1_irreps), parse_blocks(out_irreps)
    ic, oc = class_columns(ib), class_columns(ob)
    n_in = {c: len(ic[c]) for c in range(6)}
    res = {}
    for c3 in range(6):
        M = len(oc[c3])
        K = sum(n_in[c1] for c1, _, _ in paths(c3, n_in, lmax_sh))
        res[CLASSES[c3]] = (K, M) if (K > 0 and M > 0) else None
    return res


def default_norms(in1_irreps, out_irreps, lmax_sh):
    sh = shapes(in1_irreps, out_irreps, lmax_sh)
    oc = class_columns(parse_blocks(out_irreps))
    out = {}
    for c3, name in enumerate(CLASSES):
        l = c3 >> 1
        K = sh[name][0] if sh[name] else 0
        out[name] = np.full(len(oc[c3]) * (2 * l + 1), np.float32(np.sqrt((2 * l + 1) / K) if K else 1.0))
    return out


def forward(in1_irreps, out_irreps, lmax_sh, in1, in2, W, norms):
    ib, ob = parse_blocks(in1_irreps), parse_blocks(out_irreps)
    ic, oc = class_columns(ib), class_columns(ob)
    n_in = {c: len(ic[c]) for c in range(6)}
    in1 = np.asarray(in1, np.float64)
    in2 = np.asarray(in2, np.float64)
    B = in1.shape[0]
    in2 = np.broadcast_to(in2, (B, in2.shape[1]))
    out = np.zeros((B, sum((2 * l + 1) * m for l, _, m in ob)))
    for c3, name in enumerate(CLASSES):
        if len(oc[c3]) == 0:
            continue
        l3 = c3 >> 1
        feats = []
        for c1, l1, l2 in paths(c3, n_in, lmax_sh):
            x = in1[:, ic[c1]]                                    # [B, n, 2l1+1]
            y = in2[:, l2 * l2:(l2 + 1) * (l2 + 1)]               # [B, 2l2+1]
            z = np.einsum("bn,mnq->bmq", y, CG.cg(l1, l2, l3))   # [B, 2l1+1, 2l3+1]
            feats.append(np.matmul(x, z))
        if not feats:
            continue
        F = np.concatenate(feats, 1)                              # [B, K, 2l3+1]
        o = np.einsum("bkq,kw->bwq", F, np.asarray(W[name], np.float64), optimize=True)
        o = o.reshape(B, o.shape[1] * o.shape[2]) * np.asarray(norms[name], np.float64)  # explicit width: B may be 0
        out[:, oc[c3].reshape(-1)] = o
    return out


def forward_torch_cpu(in1_irreps, out_irreps, lmax_sh, in1, in2, W, norms, _cache={}):
    """fp32 torch-CPU port of the same contraction with the reference's op *pattern* (class gather ->
    per-path feature -> cat -> matmul -> column scatter -> norm); bench.py cpu_baseline for l_max = 2."""
    import torch
    key = (str(in1_irreps), str(out_irreps), lmax_sh)
    if key not in _cache:
        ib, ob = parse_blocks(in1_irreps), parse_blocks(out_irreps)
        ic, oc = class_columns(ib), class_columns(ob)
        _cache[key] = (ic, oc, {c: len(ic[c]) for c in range(6)}, sum((2 * l + 1) * m for l, _, m in ob))
    ic, oc, n_in, dout = _cache[key]
    B = in1.shape[0]
    out = torch.empty((B, dout), dtype=in1.dtype)
    for c3, name in enumerate(CLASSES):
        if len(oc[c3]) == 0:
            continue
        l3 = c3 >> 1
        feats = []
        for c1, l1, l2 in paths(c3, n_in, lmax_sh):
            x = in1[:, torch.as_tensor(ic[c1].reshape(-1))].reshape(B, -1, 2 * l1 + 1)
            y = in2[:, l2 * l2:(l2 + 1) * (l2 + 1)]
            C = torch.as_tensor(CG.cg(l1, l2, l3), dtype=in1.dtype)
            z = torch.einsum("bn,mnq->bmq", y, C)
            feats.append(torch.bmm(x, z))
        F = torch.cat(feats, 1)
        o = torch.tensordot(F, W[name], ([1], [0])).transpose(-1, -2).reshape(B, -1)
        out[:, torch.as_tensor(oc[c3].reshape(-1))] = o * norms[name]
    return out


def forward_torch_cpu_fast(in1_irreps, out_irreps, lmax_sh, in1, in2, W, norms, _cache={}):
    """Best-effort CPU variant (SURVEY.md §8d asks for one beside the faithful port, so that GPU / CPU ratios are not
    quoted against a strawman): the same arithmetic reorganised the way the GPU kernel does it -- contiguous class slices
    (no boolean-mask gathers, no cat, no column scatter), the weight contraction on the RAW channels as one GEMM per path
    (u = x W), then a small einsum with the coupling z = C . Y, written into a preallocated output view."""
    import torch
    key = (str(in1_irreps), str(out_irreps), lmax_sh)
    if key not in _cache:
        ib, ob = parse_blocks(in1_irreps), parse_blocks(out_irreps)
        ic, oc = class_columns(ib), class_columns(ob)
        n_in = {c: len(ic[c]) for c in range(6)}
        plan = []
        for c3, name in enumerate(CLASSES):
            if len(oc[c3]) == 0:
                continue
            l3, row, items = c3 >> 1, 0, []
            for c1, l1, l2 in paths(c3, n_in, lmax_sh):
                items.append((c1, l1, l2, row, n_in[c1], torch.as_tensor(CG.cg(l1, l2, l3), dtype=torch.float32)))
                row += n_in[c1]
            plan.append((c3, name, l3, items))
        _cache[key] = (ic, oc, plan, sum((2 * l + 1) * m for l, _, m in ob))
    ic, oc, plan, dout = _cache[key]
    B = in1.shape[0]
    out = torch.empty((B, dout), dtype=in1.dtype)
    xs = {}
    for c3, name, l3, items in plan:
        M = len(oc[c3])
        acc = torch.zeros((B, M, 2 * l3 + 1), dtype=in1.dtype)
        for c1, l1, l2, row, n, C in items:
            if c1 not in xs:
                cols = torch.as_tensor(ic[c1].reshape(-1))
                contiguous = bool((cols[1:] - cols[:-1] == 1).all()) if len(cols) > 1 else True
                x = in1[:, int(cols[0]):int(cols[-1]) + 1] if contiguous else in1[:, cols]
                xs[c1] = x.reshape(B, n, 2 * l1 + 1).transpose(1, 2)        # [B, 2l1+1, n] view
            u = torch.matmul(xs[c1], W[name][row:row + n])                    # [B, 2l1+1, M]  (GEMM on raw channels)
            z = torch.einsum("bn,mnq->bmq", in2[:, l2 * l2:(l2 + 1) * (l2 + 1)], C.to(in1.dtype))   # [B, 2l1+1, 2l3+1]
            acc += torch.einsum("bam,baq->bmq", u, z)
        out[:, torch.as_tensor(oc[c3].reshape(-1))] = acc.reshape(B, -1) * norms[name]
    return out
