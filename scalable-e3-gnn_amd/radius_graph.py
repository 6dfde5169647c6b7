"""Radius graph on the GPU (host side of ``e3_rg_*`` in include/e3gnn.h).

Builder-defined stage (no reference code in the mount): particles are renumbered by
a stable Morton sort, edges ``(src=j -> dst=i)`` exist iff ``i != j`` and ``|x_i-x_j|^2 <= r^2`` in
explicitly rounded fp32, and the result is CSR-by-dst with ascending ``src``.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass

import torch

from . import _lib


class RgParams(ctypes.Structure):
    _fields_ = [("lo", ctypes.c_float * 3), ("hi", ctypes.c_float * 3), ("r", ctypes.c_float),
                ("n", ctypes.c_int32 * 3), ("inv", ctypes.c_float * 3), ("bits", ctypes.c_int32)]


@dataclass
class RadiusGraph:
    perm: torch.Tensor      # [N] int32   new id -> original id
    pos4: torch.Tensor      # [N,4] fp32  positions in new order (x,y,z,0)
    rowptr: torch.Tensor    # [N+1] int32 CSR by dst
    src: torch.Tensor       # [E] int32   ascending inside each row
    num_edges: int
    grid: tuple

    @property
    def dst(self) -> torch.Tensor:
        """[E] int32 destination of every edge (expanded from rowptr once, then cached)."""
        d = getattr(self, "_dst", None)
        if d is None:
            n = self.rowptr.numel() - 1
            deg = (self.rowptr[1:] - self.rowptr[:-1]).long()
            d = torch.repeat_interleave(torch.arange(n, device=self.src.device, dtype=torch.int32), deg,
                                        output_size=self.num_edges)
            object.__setattr__(self, "_dst", d)
        return d


def grid_params(lo, hi, r) -> RgParams:
    p = RgParams()
    for a in range(3):
        p.lo[a], p.hi[a] = float(lo[a]), float(hi[a])
    p.r = float(r)
    _lib.check(_lib.load().e3_rg_grid(ctypes.byref(p)), "e3_rg_grid")
    return p


def radius_graph(pos: torch.Tensor, r: float, lo=None, hi=None) -> RadiusGraph:
    """pos [N,3] fp32 on a ROCm device.  ``lo``/``hi``: bounding box (computed from pos when omitted)."""
    if not pos.is_cuda:
        raise RuntimeError("radius_graph runs on ROCm tensors only; there is no CPU path")
    if pos.dtype != torch.float32 or pos.dim() != 2 or pos.shape[1] != 3:
        raise RuntimeError(f"pos must be [N,3] float32, got {tuple(pos.shape)} {pos.dtype}")
    pos = pos.contiguous()
    N = pos.shape[0]
    if lo is None or hi is None:
        lo = pos.min(0).values.tolist() if N else [0.0, 0.0, 0.0]
        hi = pos.max(0).values.tolist() if N else [1.0, 1.0, 1.0]
        hi = [h if h > l else l + 1.0 for l, h in zip(lo, hi)]
    lib = _lib.load()
    p = grid_params(lo, hi, r)
    dev = pos.device
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        wbytes = lib.e3_rg_workspace_bytes(N, ctypes.byref(p))
        if wbytes < 0:
            raise RuntimeError("e3_rg_workspace_bytes: invalid arguments")
        ws = torch.empty(max(int(wbytes), 16), dtype=torch.uint8, device=dev)
        perm = torch.empty(N, dtype=torch.int32, device=dev)
        pos4 = torch.empty((N, 4), dtype=torch.float32, device=dev)
        rowptr = torch.empty(N + 1, dtype=torch.int32, device=dev)
        _lib.check(lib.e3_rg_sort_count(pos.data_ptr(), N, ctypes.byref(p), perm.data_ptr(), pos4.data_ptr(),
                                        rowptr.data_ptr(), ws.data_ptr(), wbytes, stream), "e3_rg_sort_count")
        # the count / scan run in int32 (indices are int32 end to end): a graph with >= 2^31 edges wraps the running sum,
        # which shows as a negative or decreasing rowptr -- checked here with the same host read that fetches E
        if N > 0:
            E, mindeg = torch.stack([rowptr[-1], (rowptr[1:] - rowptr[:-1]).min()]).tolist()
        else:
            E, mindeg = int(rowptr[-1].item()), 0
        if E < 0 or mindeg < 0:
            raise RuntimeError("radius_graph: the edge count does not fit int32 (>= 2^31 edges); shard the cloud "
                               "(sharding.SlabHalo) or reduce the cutoff")
        E = int(E)
        src = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
        _lib.check(lib.e3_rg_fill(N, ctypes.byref(p), pos4.data_ptr(), rowptr.data_ptr(), src.data_ptr(),
                                  ws.data_ptr(), wbytes, stream), "e3_rg_fill")
    return RadiusGraph(perm, pos4, rowptr, src[:E], E, (tuple(p.n), p.bits))
