"""Spatial sharding of the point cloud across GPUs (one process per GPU, RCCL over xGMI).

Builder-defined (the reference has no distributed code).  The domain is cut into a grid of axis-aligned
boxes, one per rank (``GridHalo``); a node's messages need neighbours within the cutoff ``r``, so each rank also holds
*ghost* copies of the particles of the (up to 26) adjacent boxes that lie within ``r`` of its own box:

  * ``setup``      — once per graph build: boundary particles (positions + input features) go to the adjacent boxes; the
                     local cloud is ``[owned | ghosts from neighbour 0 | ghosts from neighbour 1 | ...]``.
  * ``split_graph``— once per graph build: edges INTO ghost rows are dropped (their sums would be thrown away), the rest is
                     split into *interior* edges (owned src: computable before the layer's exchange has landed) and
                     *boundary* edges (ghost src).  On a GPU one HIP launch pair classifies and compacts (``e3_split_edges``).
  * ``start`` / ``finish`` — once per layer: the refreshed features of the boundary particles are posted (grouped
                     isend/irecv), the interior edges' message kernel runs meanwhile, ``finish`` waits and writes the ghost
                     rows of ``h`` IN PLACE (inference only: see ``finish``), then the boundary edges run.

Two layouts are built on it:

  * ``SlabHalo``   — slabs along x (grid N x 1 x 1): what ``bench.py --gpus N`` uses for WEAK scaling (N unit cubes side by
                     side, 1 M particles each): a face costs r / 1 = 1.8 % ghosts per side, every rank has <= 2 neighbours.
  * ``GridHalo((2, 2, 2), ...)`` — octants of ONE box, i.e. the top level of the Morton order (each octant is one contiguous
                     Morton key range): the STRONG-scaling layout.  A unit cube cut 8 ways costs ~3 r / (1/2)
                     = 11 % ghosts per rank spread over 7 neighbours (7 xGMI links) against 2 r / (1/8) = 29 % over 2 links
                     for slabs of width 1/8; tests/test_sharding_gloo.py prints both.  (Equal-COUNT Morton ranges for a
                     non-uniform cloud are not implemented: boxes are equal-volume.)

Only point-to-point traffic between adjacent boxes (``batch_isend_irecv`` = grouped ncclSend/ncclRecv on RCCL: every pair
talks over its own xGMI link; no ring, no collective over all ranks).  Host syncs: two per graph build (one ``nonzero`` over
all neighbours at once, one read of the incoming counts), one per graph in ``split_graph`` (three edge counts), none per layer.
"""
from __future__ import annotations

import ctypes
import itertools
from dataclasses import dataclass

import torch
import torch.distributed as dist


@dataclass
class SplitGraph:
    """Edge lists of a sharded graph, all sorted by dst (CSR order) in the graph's local (Morton) numbering."""
    graph: object                 # RadiusGraph with the edges into ghost rows removed (rowptr / src / dst consistent)
    interior: tuple               # (src int32 [Ei], dst int32 [Ei]): owned src
    boundary: tuple               # (src int32 [Eb], dst int32 [Eb]): ghost src
    dropped: int                  # edges into ghost rows that were removed


class GridHalo:
    """Ghost-cell halo of a ``dims = (px, py, pz)`` grid of equal boxes covering ``[lo, hi)``; rank = (ix py + iy) pz + iz."""

    def __init__(self, dims, lo, hi, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.dims = tuple(int(d) for d in dims)
        if self.dims[0] * self.dims[1] * self.dims[2] != self.world:
            raise ValueError(f"grid {self.dims} needs {self.dims[0] * self.dims[1] * self.dims[2]} ranks, the group has {self.world}")
        self.lo = [float(v) for v in lo]
        self.hi = [float(v) for v in hi]
        self.n_owned = 0
        self.bytes_last_exchange = 0
        self.neighbours = []       # adjacent ranks, ascending
        self._set_boxes()

    # -- geometry -------------------------------------------------------------------------------------
    def coords(self, rank):
        px, py, pz = self.dims
        return rank // (py * pz), (rank // pz) % py, rank % pz

    def box(self, rank):
        c = self.coords(rank)
        w = [(self.hi[a] - self.lo[a]) / self.dims[a] for a in range(3)]
        return [self.lo[a] + c[a] * w[a] for a in range(3)], [self.lo[a] + (c[a] + 1) * w[a] for a in range(3)]

    def owner_of(self, pos: torch.Tensor) -> torch.Tensor:
        """Rank that owns each position (positions outside [lo, hi) are clamped into the edge boxes)."""
        idx = []
        for a in range(3):
            w = (self.hi[a] - self.lo[a]) / self.dims[a]
            idx.append(((pos[:, a] - self.lo[a]) / w).floor().long().clamp_(0, self.dims[a] - 1))
        return (idx[0] * self.dims[1] + idx[1]) * self.dims[2] + idx[2]

    def _set_boxes(self):
        me = self.coords(self.rank)
        nb = set()
        for d in itertools.product((-1, 0, 1), repeat=3):
            c = [me[a] + d[a] for a in range(3)]
            if d != (0, 0, 0) and all(0 <= c[a] < self.dims[a] for a in range(3)):
                nb.add((c[0] * self.dims[1] + c[1]) * self.dims[2] + c[2])
        self.neighbours = sorted(nb)

    # -- p2p helpers ----------------------------------------------------------------------------------
    def _staged(self, t):
        """gloo has no device transport: device tensors are staged through host memory (rehearsal mode only)."""
        return t.is_cuda and dist.get_backend(self.group) == "gloo"

    def _post(self, sends, recvs):
        """Grouped send / recv with every neighbour (empty messages are skipped on both ends: the sizes were agreed on in
        ``setup``)."""
        ops = []
        for q, s, r in zip(self.neighbours, sends, recvs):
            if s.numel():
                ops.append(dist.P2POp(dist.isend, s, q, self.group))
            if r.numel():
                ops.append(dist.P2POp(dist.irecv, r, q, self.group))
        return dist.batch_isend_irecv(ops) if ops else []

    def _sendrecv(self, sends, recvs):
        if sends and self._staged(sends[0]):
            hs, hr = [t.cpu() for t in sends], [t.cpu() for t in recvs]
            for w in self._post(hs, hr):
                w.wait()
            for d, s in zip(recvs, hr):
                d.copy_(s)
            return
        for w in self._post(sends, recvs):
            w.wait()

    # -- once per graph build -------------------------------------------------------------------------
    def setup(self, pos: torch.Tensor, feats: torch.Tensor, r: float):
        """pos [n,3], feats [n,F] of the owned particles -> (local_pos, local_feats) with the ghosts of every adjacent box
        appended in neighbour order.  Positions and features travel in their own dtypes."""
        dev = pos.device
        n = pos.shape[0]
        self.n_owned = n
        nn = len(self.neighbours)
        w = min((self.hi[a] - self.lo[a]) / self.dims[a] for a in range(3) if self.dims[a] > 1) if nn else float("inf")
        if nn and r > w:
            raise ValueError(f"cutoff {r} exceeds the box width {w}: ghosts would come from beyond the adjacent boxes")
        if nn:
            # my particles within r (per axis: a superset of the r-ball) of each adjacent box -- one mask, ONE nonzero
            blo = torch.tensor([self.box(q)[0] for q in self.neighbours], dtype=pos.dtype, device=dev)   # [nn, 3]
            bhi = torch.tensor([self.box(q)[1] for q in self.neighbours], dtype=pos.dtype, device=dev)
            m = ((pos[None, :, :] >= blo[:, None, :] - r) & (pos[None, :, :] < bhi[:, None, :] + r)).all(-1)   # [nn, n]
            nbr, idx = m.nonzero(as_tuple=True)                     # sorted by neighbour, then particle  (host sync #1)
            cnt_out = torch.bincount(nbr, minlength=nn)
        else:
            idx = torch.empty(0, dtype=torch.long, device=dev)
            cnt_out = torch.zeros(0, dtype=torch.int64, device=dev)
        cnt_in = torch.zeros_like(cnt_out)
        self._sendrecv([cnt_out[i:i + 1] for i in range(nn)], [cnt_in[i:i + 1] for i in range(nn)])
        both = torch.stack([cnt_out, cnt_in]).tolist() if nn else [[], []]                               # host sync #2
        self.send_counts, self.recv_counts = [int(v) for v in both[0]], [int(v) for v in both[1]]
        self.sel = idx                                   # original indices of the particles sent, grouped by neighbour
        self._send_idx = idx
        out = []
        ng = sum(self.recv_counts)
        for t in (pos, feats):
            sends = list(t[idx].contiguous().split(self.send_counts)) if nn else []
            ghosts = torch.empty((ng, t.shape[1]), dtype=t.dtype, device=dev)
            self._sendrecv(sends, list(ghosts.split(self.recv_counts)) if nn else [])
            out.append(torch.cat([t, ghosts], 0))
        self.n_ghost = ng
        self._recv_idx = torch.arange(n, n + ng, device=dev)
        return out[0], out[1]

    def renumber(self, perm: torch.Tensor):
        """The graph builder renumbers the local cloud (``perm[new] = old``): translate the halo index lists."""
        inv = torch.empty_like(perm, dtype=torch.long)
        inv[perm.long()] = torch.arange(perm.numel(), device=perm.device)
        n = self.n_owned
        self._send_idx = inv[self.sel]
        self._recv_idx = inv[n:n + self.n_ghost]   # new ids of the ghosts, arrival order
        self.owned_new = inv[:n]                    # new ids of the owned particles, in their original order
        self.is_ghost = torch.zeros(perm.numel(), dtype=torch.bool, device=perm.device)
        self.is_ghost[self._recv_idx] = True
        return self

    def ghost_fraction(self) -> float:
        return self.n_ghost / max(1, self.n_owned)

    def split_graph(self, g) -> SplitGraph:
        """Drop the edges into ghost rows and split the rest by the ownership of their src (see the module docstring).
        ``g``: the RadiusGraph of the local cloud (after ``renumber(g.perm)``)."""
        from .radius_graph import RadiusGraph
        src, dst = g.src, g.dst
        N, E = g.rowptr.numel() - 1, int(src.numel())
        if src.is_cuda:
            # one classification + compaction on the device (csrc/e3_shard.hip); ONE host read for the three counts
            from . import _lib
            lib = _lib.load()
            dev = src.device
            ghost8 = self.is_ghost.to(torch.uint8)
            rowptr2 = torch.empty(N + 1, dtype=torch.int32, device=dev)
            outs = [torch.empty(max(E, 1), dtype=torch.int32, device=dev) for _ in range(6)]
            counts = torch.empty(4, dtype=torch.int32, device=dev)
            work = torch.empty(int(lib.e3_split_edges_workspace_bytes(N)), dtype=torch.uint8, device=dev)
            with torch.cuda.device(dev):
                _lib.check(lib.e3_split_edges(g.rowptr.data_ptr(), src.data_ptr(), ghost8.data_ptr(), N, E,
                                              rowptr2.data_ptr(), *[o.data_ptr() for o in outs], counts.data_ptr(),
                                              work.data_ptr(), torch.cuda.current_stream(dev).cuda_stream), "e3_split_edges")
            ek, ei, eb = (int(v) for v in counts[:3].tolist())
            g2 = RadiusGraph(g.perm, g.pos4, rowptr2, outs[0][:ek], ek, g.grid)
            object.__setattr__(g2, "_dst", outs[1][:ek])
            return SplitGraph(g2, (outs[2][:ei], outs[3][:ei]), (outs[4][:eb], outs[5][:eb]), E - ek)
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

    # -- once per layer -------------------------------------------------------------------------------
    def start(self, h: torch.Tensor):
        """Post this layer's ghost refresh (boundary rows of ``h`` to the neighbours) and return a token for ``finish``.
        Kernels launched between the two calls overlap the transfer as long as they do not read ghost rows."""
        D = h.shape[1]
        recv = torch.empty((self.n_ghost, D), dtype=h.dtype, device=h.device)
        send = h[self._send_idx].contiguous()
        self.bytes_last_exchange = send.numel() * h.element_size()
        sends, recvs = list(send.split(self.send_counts)), list(recv.split(self.recv_counts))
        if self._staged(h):
            hs, hr = [t.cpu() for t in sends], [t.cpu() for t in recvs]
            return ("staged", self._post(hs, hr), recv, (hs, hr, recvs))
        return ("direct", self._post(sends, recvs), recv, (send,))

    def finish(self, h: torch.Tensor, token) -> torch.Tensor:
        """Wait for the transfer and write the ghost rows of ``h`` in place (one indexed copy, no clone of ``h``).
        Inference only: a tensor autograd has saved for backward must not be overwritten."""
        if torch.is_grad_enabled() and h.requires_grad:
            raise RuntimeError("the halo refresh writes ghost rows in place: not differentiable (run under torch.no_grad())")
        kind, works, recv, keep = token
        for w in works:
            w.wait()
        if kind == "staged":
            for d, s in zip(keep[2], keep[1]):
                d.copy_(s)
        if recv.shape[0]:
            h.index_copy_(0, self._recv_idx, recv)
        return h

    def exchange(self, h: torch.Tensor) -> torch.Tensor:
        """Blocking form: overwrite the ghost rows of ``h`` (local numbering) with the owners' current values, in place."""
        return self.finish(h, self.start(h))

    def ghost_rows(self) -> torch.Tensor:
        """Local (graph-order) ids of the ghost rows."""
        return self._recv_idx


class SlabHalo(GridHalo):
    """Slabs along x: rank k owns ``x in [slab_lo, slab_hi)`` (given at ``setup``), unbounded in y and z."""

    def __init__(self, group=None):
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        super().__init__((world, 1, 1), (0.0, -1e30, -1e30), (float(world), 1e30, 1e30), group)
        self.left = self.rank - 1 if self.rank > 0 else None
        self.right = self.rank + 1 if self.rank < self.world - 1 else None

    def setup(self, pos, feats, slab_lo: float, slab_hi: float, r: float):
        w = float(slab_hi) - float(slab_lo)
        self.lo[0] = float(slab_lo) - self.rank * w
        self.hi[0] = self.lo[0] + self.world * w
        out = super().setup(pos, feats, r)
        by = dict(zip(self.neighbours, self.recv_counts))
        self.n_ghost_left = by.get(self.left, 0) if self.left is not None else 0
        self.n_ghost_right = by.get(self.right, 0) if self.right is not None else 0
        return out
