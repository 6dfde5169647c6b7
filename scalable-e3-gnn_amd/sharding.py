"""Spatial sharding of the point cloud across GPUs (one process per GPU, RCCL over xGMI).

Builder-defined (the reference has no distributed code).  The cloud is cut into
slabs along x; rank k owns ``x in [lo_k, hi_k)``.  A node's messages need neighbours within the cutoff
``r``, so each rank also holds *ghost* copies of the neighbouring slabs' particles within ``r`` of its
faces:

  * ``setup``      — once per graph build: boundary particles (positions + input features) go to the two
                     slab neighbours; the local cloud is ``[owned | ghosts from left | ghosts from right]``.
  * ``split_graph``— once per graph build: edges INTO ghost rows are dropped (their sums would be thrown
                     away), the rest is split into *interior* edges (owned src: computable before the
                     layer's exchange has landed) and *boundary* edges (ghost src).
  * ``start`` / ``finish`` — once per layer: the refreshed features of the boundary particles are posted
                     (grouped isend/irecv), the interior edges' message kernel runs meanwhile, ``finish``
                     waits and writes the ghost rows of ``h`` IN PLACE, then the boundary edges run.

Only point-to-point traffic between slab neighbours (``batch_isend_irecv`` = grouped ncclSend/ncclRecv on
RCCL: every pair talks over its own xGMI link; no ring, no collective over all ranks).  Pure
``torch`` + ``torch.distributed``: device-agnostic, so the same code runs under gloo on CPU in the tests.

Why slabs (and when not): `bench.py --gpus N` scales WEAKLY along x (N unit cubes side by side, 1 M particles
each), where a slab face costs r / 1 = 1.8 % ghosts per side (3.6 % for a middle rank) and every rank has at most two
neighbours, each on its own xGMI link.  For a FIXED unit cube cut 8 ways, slabs of width 1/8 pay 2 r / (1/8) = 14 % ghosts
on 2 links, a 2 x 2 x 2 Morton-range (octant) partition 3 r / (1/2) + edges = 5.5 % on 7 links: that
strong-scaling layout is the better one there and is not implemented.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch
import torch.distributed as dist


@dataclass
class SplitGraph:
    """Edge lists of a sharded graph, all sorted by dst (CSR order) in the graph's local (Morton) numbering."""
    graph: object                 # RadiusGraph with the edges into ghost rows removed (rowptr / src / dst consistent)
    interior: tuple               # (src int32 [Ei], dst int32 [Ei]): owned src
    boundary: tuple               # (src int32 [Eb], dst int32 [Eb]): ghost src
    dropped: int                  # edges into ghost rows that were removed


class SlabHalo:
    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.left = self.rank - 1 if self.rank > 0 else None
        self.right = self.rank + 1 if self.rank < self.world - 1 else None
        self.n_owned = 0
        self.bytes_last_exchange = 0

    # -- p2p helpers --------------------------------------------------------------------------------
    def _post(self, to_left, to_right, from_left, from_right):
        """Post the grouped send/recv; returns the work handles (empty messages are skipped on both ends: the sizes
        were agreed on in ``setup``)."""
        ops = []
        if self.left is not None:
            if to_left.numel():
                ops.append(dist.P2POp(dist.isend, to_left, self.left, self.group))
            if from_left.numel():
                ops.append(dist.P2POp(dist.irecv, from_left, self.left, self.group))
        if self.right is not None:
            if to_right.numel():
                ops.append(dist.P2POp(dist.isend, to_right, self.right, self.group))
            if from_right.numel():
                ops.append(dist.P2POp(dist.irecv, from_right, self.right, self.group))
        return dist.batch_isend_irecv(ops) if ops else []

    def _staged(self, t):
        """gloo has no device transport: device tensors are staged through host memory (rehearsal mode only)."""
        return t.is_cuda and dist.get_backend(self.group) == "gloo"

    def _sendrecv(self, to_left, to_right, from_left, from_right):
        if self._staged(to_left):
            bufs = [t.cpu() for t in (to_left, to_right, from_left, from_right)]
            for w in self._post(*bufs):
                w.wait()
            from_left.copy_(bufs[2])
            from_right.copy_(bufs[3])
            return
        for w in self._post(to_left, to_right, from_left, from_right):
            w.wait()

    # -- once per graph build -----------------------------------------------------------------------
    def setup(self, pos: torch.Tensor, feats: torch.Tensor, slab_lo: float, slab_hi: float, r: float):
        """pos [n,3], feats [n,F] of the owned particles -> (local_pos, local_feats) with ghosts appended.
        Positions and features travel in their own dtypes (two messages per neighbour), one host read for the counts."""
        dev = pos.device
        n = pos.shape[0]
        self.n_owned = n
        empty = torch.empty(0, dtype=torch.long, device=dev)
        self.sel_left = (pos[:, 0] < slab_lo + r).nonzero().flatten() if self.left is not None else empty
        self.sel_right = (pos[:, 0] >= slab_hi - r).nonzero().flatten() if self.right is not None else empty
        cnt_out = torch.tensor([self.sel_left.numel(), self.sel_right.numel()], dtype=torch.int64, device=dev)
        cnt_in = torch.zeros(2, dtype=torch.int64, device=dev)
        self._sendrecv(cnt_out[0:1], cnt_out[1:2], cnt_in[0:1], cnt_in[1:2])
        self.n_ghost_left, self.n_ghost_right = (int(v) for v in cnt_in.tolist())
        out = []
        for t in (pos, feats):
            gl = torch.empty((self.n_ghost_left, t.shape[1]), dtype=t.dtype, device=dev)
            gr = torch.empty((self.n_ghost_right, t.shape[1]), dtype=t.dtype, device=dev)
            self._sendrecv(t[self.sel_left].contiguous(), t[self.sel_right].contiguous(), gl, gr)
            out.append(torch.cat([t, gl, gr], 0))
        self._send_left_idx = self.sel_left
        self._send_right_idx = self.sel_right
        self._recv_idx = torch.arange(n, n + self.n_ghost_left + self.n_ghost_right, device=dev)
        return out[0], out[1]

    def renumber(self, perm: torch.Tensor):
        """The graph builder renumbers the local cloud (``perm[new] = old``): translate the halo index lists."""
        inv = torch.empty_like(perm, dtype=torch.long)
        inv[perm.long()] = torch.arange(perm.numel(), device=perm.device)
        self._send_left_idx = inv[self.sel_left]
        self._send_right_idx = inv[self.sel_right]
        n, gl, gr = self.n_owned, self.n_ghost_left, self.n_ghost_right
        self._recv_idx = inv[n:n + gl + gr]      # new ids of [ghosts from left | ghosts from right], arrival order
        self.owned_new = inv[:n]                 # new ids of the owned particles, in their original order
        self.is_ghost = torch.zeros(perm.numel(), dtype=torch.bool, device=perm.device)
        self.is_ghost[self._recv_idx] = True
        return self

    def split_graph(self, g) -> SplitGraph:
        """Drop the edges into ghost rows and split the rest by the ownership of their src (see the module docstring).
        ``g``: the RadiusGraph of the local cloud (after ``renumber(g.perm)``)."""
        from .radius_graph import RadiusGraph
        src, dst = g.src, g.dst
        keep = ~self.is_ghost[dst.long()]
        src_k, dst_k = src[keep], dst[keep]
        deg = (g.rowptr[1:] - g.rowptr[:-1]).clone()
        deg[self.is_ghost] = 0
        rowptr = torch.zeros_like(g.rowptr)
        rowptr[1:] = torch.cumsum(deg, 0)
        g2 = RadiusGraph(g.perm, g.pos4, rowptr, src_k.contiguous(), int(src_k.numel()), g.grid)
        object.__setattr__(g2, "_dst", dst_k.contiguous())
        ghost_src = self.is_ghost[src_k.long()]
        interior = (src_k[~ghost_src].contiguous(), dst_k[~ghost_src].contiguous())
        boundary = (src_k[ghost_src].contiguous(), dst_k[ghost_src].contiguous())
        return SplitGraph(g2, interior, boundary, int(src.numel() - src_k.numel()))

    # -- once per layer -----------------------------------------------------------------------------
    def start(self, h: torch.Tensor):
        """Post this layer's ghost refresh (boundary rows of ``h`` to the neighbours) and return a token for ``finish``.
        Kernels launched between the two calls overlap the transfer as long as they do not read ghost rows."""
        D = h.shape[1]
        recv = torch.empty((self.n_ghost_left + self.n_ghost_right, D), dtype=h.dtype, device=h.device)
        gl, gr = recv[:self.n_ghost_left], recv[self.n_ghost_left:]
        sl, sr = h[self._send_left_idx].contiguous(), h[self._send_right_idx].contiguous()
        self.bytes_last_exchange = (sl.numel() + sr.numel()) * h.element_size()
        if self._staged(h):
            bufs = [t.cpu() for t in (sl, sr, gl, gr)]
            return ("staged", self._post(*bufs), recv, bufs)
        return ("direct", self._post(sl, sr, gl, gr), recv, (sl, sr))

    def finish(self, h: torch.Tensor, token) -> torch.Tensor:
        """Wait for the transfer and write the ghost rows of ``h`` in place (one indexed copy, no clone of ``h``)."""
        kind, works, recv, keep = token
        for w in works:
            w.wait()
        if kind == "staged":
            recv[:self.n_ghost_left].copy_(keep[2])
            recv[self.n_ghost_left:].copy_(keep[3])
        if recv.shape[0]:
            h.index_copy_(0, self._recv_idx, recv)
        return h

    def exchange(self, h: torch.Tensor) -> torch.Tensor:
        """Blocking form: overwrite the ghost rows of ``h`` (local numbering) with the owners' current values, in place."""
        return self.finish(h, self.start(h))
