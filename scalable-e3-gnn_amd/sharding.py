"""Spatial sharding of the point cloud across GPUs (one process per GPU, RCCL over xGMI).

Builder-defined (the reference has no distributed code).  The cloud is cut into
slabs along x; rank k owns ``x in [lo_k, hi_k)``.  A node's messages need neighbours within the cutoff
``r``, so each rank also holds *ghost* copies of the neighbouring slabs' particles within ``r`` of its
faces:

  * ``setup``    — once per graph build: boundary particles (positions + input features) go to the two
                   slab neighbours; the local cloud is ``[owned | ghosts from left | ghosts from right]``.
  * ``exchange`` — once per layer: refreshed features of the boundary particles overwrite the ghost rows.

Only point-to-point traffic between slab neighbours (``batch_isend_irecv`` = grouped ncclSend/ncclRecv on
RCCL: every pair talks over its own xGMI link; no ring, no collective over all ranks).  Pure
``torch`` + ``torch.distributed``: device-agnostic, so the same code runs under gloo on CPU in the tests.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist


class SlabHalo:
    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.left = self.rank - 1 if self.rank > 0 else None
        self.right = self.rank + 1 if self.rank < self.world - 1 else None
        self.n_owned = 0
        self.bytes_last_exchange = 0

    # -- p2p helper ---------------------------------------------------------------------------------
    def _sendrecv(self, to_left, to_right, from_left, from_right):
        if to_left.is_cuda and dist.get_backend(self.group) == "gloo":
            # rehearsal mode (gloo has no device transport): stage through host memory
            bufs = [t.cpu() for t in (to_left, to_right, from_left, from_right)]
            self._sendrecv(*bufs)
            from_left.copy_(bufs[2])
            from_right.copy_(bufs[3])
            return
        ops = []  # empty messages are skipped on both ends (sizes were agreed on in `setup`)
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
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()

    # -- once per graph build -----------------------------------------------------------------------
    def setup(self, pos: torch.Tensor, feats: torch.Tensor, slab_lo: float, slab_hi: float, r: float):
        """pos [n,3], feats [n,F] of the owned particles -> (local_pos, local_feats) with ghosts appended."""
        dev = pos.device
        n = pos.shape[0]
        self.n_owned = n
        empty = torch.empty(0, dtype=torch.long, device=dev)
        self.sel_left = (pos[:, 0] < slab_lo + r).nonzero().flatten() if self.left is not None else empty
        self.sel_right = (pos[:, 0] >= slab_hi - r).nonzero().flatten() if self.right is not None else empty
        cnt_out = torch.tensor([self.sel_left.numel(), self.sel_right.numel()], dtype=torch.int64, device=dev)
        cnt_l, cnt_r = torch.zeros(1, dtype=torch.int64, device=dev), torch.zeros(1, dtype=torch.int64, device=dev)
        self._sendrecv(cnt_out[0:1].contiguous(), cnt_out[1:2].contiguous(), cnt_l, cnt_r)
        self.n_ghost_left, self.n_ghost_right = int(cnt_l.item()), int(cnt_r.item())
        payload = torch.cat([pos, feats.to(pos.dtype)], 1)  # one message; bf16 features ride as fp32 (lossless)
        F = payload.shape[1]
        gl = torch.empty((self.n_ghost_left, F), dtype=payload.dtype, device=dev)
        gr = torch.empty((self.n_ghost_right, F), dtype=payload.dtype, device=dev)
        self._sendrecv(payload[self.sel_left].contiguous(), payload[self.sel_right].contiguous(), gl, gr)
        local = torch.cat([payload, gl, gr], 0)
        self._send_left_idx = self.sel_left
        self._send_right_idx = self.sel_right
        self._recv_left_idx = torch.arange(n, n + self.n_ghost_left, device=dev)
        self._recv_right_idx = torch.arange(n + self.n_ghost_left, n + self.n_ghost_left + self.n_ghost_right, device=dev)
        return local[:, :3].contiguous(), local[:, 3:].to(feats.dtype).contiguous()

    def renumber(self, perm: torch.Tensor):
        """The graph builder renumbers the local cloud (``perm[new] = old``): translate the halo index lists."""
        inv = torch.empty_like(perm, dtype=torch.long)
        inv[perm.long()] = torch.arange(perm.numel(), device=perm.device)
        self._send_left_idx = inv[self.sel_left]
        self._send_right_idx = inv[self.sel_right]
        n, gl, gr = self.n_owned, self.n_ghost_left, self.n_ghost_right
        self._recv_left_idx = inv[n:n + gl]
        self._recv_right_idx = inv[n + gl:n + gl + gr]
        self.owned_new = inv[:n]          # new ids of the owned particles, in their original order
        return self

    # -- once per layer -----------------------------------------------------------------------------
    def exchange(self, h: torch.Tensor) -> torch.Tensor:
        """Overwrite ghost rows of ``h`` (local numbering) with the owners' current values."""
        D = h.shape[1]
        gl = torch.empty((self.n_ghost_left, D), dtype=h.dtype, device=h.device)
        gr = torch.empty((self.n_ghost_right, D), dtype=h.dtype, device=h.device)
        sl, sr = h[self._send_left_idx].contiguous(), h[self._send_right_idx].contiguous()
        self._sendrecv(sl, sr, gl, gr)
        self.bytes_last_exchange = (sl.numel() + sr.numel()) * h.element_size()
        h = h.clone()
        h[self._recv_left_idx] = gl
        h[self._recv_right_idx] = gr
        return h
