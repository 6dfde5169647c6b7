"""Small-graph / many-batch path (BASELINE.json configs[3]: QM9-style batched molecules, energy head).

No reference code exists for this stage in the mount (SURVEY.md §8 row a-N5): the contract here is builder-defined.
A batch of molecules is ONE radius graph: the molecules are laid out on a 3-D lattice whose spacing exceeds the largest
molecule by two cutoff radii, so the cell-list builder (`e3_rg_*`) finds exactly the intra-molecular pairs in one pass,
with the same kernels as the single-cloud path.  The neighbour search runs on the lattice copy; the graph that is
returned carries the ORIGINAL coordinates, so edge vectors / spherical harmonics see no shift rounding.  The energy
head is the per-molecule sum of a scalar (`1x0e`) node readout.  The force head is -dE/dpos by reverse-mode autograd
through the whole message pass: every stage has a HIP backward (tensor products: `e3_l1tp_backward` / `e3_tp_backward`;
edge geometry, gather/concat, gates, segment-sum: `e3_*_backward`, csrc/e3_edge_bwd.hip).  First order only: forces can
be predicted and energies trained on; training ON forces would need the second derivative, which is not implemented.
"""
from __future__ import annotations

import dataclasses
import math

import torch
from torch import nn

from .radius_graph import RadiusGraph, radius_graph
from .segnn import SEGNN


def batched_radius_graph(pos: torch.Tensor, batch: torch.Tensor, r: float):
    """pos [N,3] fp32 on a ROCm device, batch [N] integer molecule id of every atom (any order, ids 0..n_mol-1).
    -> (RadiusGraph over all atoms with intra-molecular edges only, mol_of_node [N] int64 in the graph's node order)"""
    if not pos.is_cuda:
        raise RuntimeError("batched_radius_graph runs on ROCm tensors only; there is no CPU path")
    batch = batch.long()
    N = pos.shape[0]
    n_mol = int(batch.max().item()) + 1 if N else 0
    if N == 0:
        g = radius_graph(pos, r, [0.0] * 3, [1.0] * 3)
        return g, batch
    idx = batch[:, None].expand(-1, 3)
    pmin = torch.full((n_mol, 3), float("inf"), device=pos.device).scatter_reduce(0, idx, pos, "amin")
    pmax = torch.full((n_mol, 3), float("-inf"), device=pos.device).scatter_reduce(0, idx, pos, "amax")
    extent = float((pmax - pmin).max().item())
    cell = extent + 2.0 * float(r)
    side = max(1, math.ceil(n_mol ** (1.0 / 3.0) - 1e-9))
    lattice = torch.stack([batch % side, (batch // side) % side, batch // (side * side)], 1).to(pos.dtype)
    shifted = (pos - pmin[batch]) + lattice * cell + float(r)
    nz = (n_mol + side * side - 1) // (side * side)
    hi = [side * cell + float(r), side * cell + float(r), nz * cell + float(r)]
    if max(hi) / float(r) > 1000:
        raise RuntimeError("batch too large for one lattice pass (cell grid > 1000 cells per axis): split the batch")
    g = radius_graph(shifted.contiguous(), r, [0.0, 0.0, 0.0], hi)
    perm = g.perm.long()
    pos4 = torch.zeros((N, 4), dtype=torch.float32, device=pos.device)
    pos4[:, :3] = pos[perm]
    return dataclasses.replace(g, pos4=pos4), batch[perm]


class BatchedEnergyModel(nn.Module):
    """SEGNN with a scalar node readout summed per molecule: energies [n_mol]."""

    def __init__(self, in_irreps="1x0e+1x1o", hidden: int = 32, num_layers: int = 4, lmax: int = 2):
        super().__init__()
        self.net = SEGNN(in_irreps, hidden, "1x0e", num_layers, lmax=lmax)

    def forward(self, x: torch.Tensor, pos: torch.Tensor, batch: torch.Tensor, r: float, forces: bool = False):
        """-> energies [n_mol]; with ``forces=True``: (energies, forces [N,3] = -dE/dpos in the caller's atom order)."""
        from . import ops
        g, mol = batched_radius_graph(pos, batch, r)
        n_mol = int(batch.max().item()) + 1 if batch.numel() else 0
        perm = g.perm.long()
        if not forces:
            e_node = self.net(x[perm], g)
            return torch.zeros(n_mol, dtype=e_node.dtype, device=e_node.device).index_add_(0, mol, e_node[:, 0])
        with torch.enable_grad():
            pos_g = pos[perm].detach().float().requires_grad_(True)       # graph (Morton) order
            geometry = ops.edge_geometry(g, lmax=self.net.lmax, pos=pos_g)  # differentiable Y, d, A
            e_node = self.net(x[perm], g, geometry=geometry)
            energy = torch.zeros(n_mol, dtype=e_node.dtype, device=e_node.device).index_add(0, mol, e_node[:, 0])
            (gpos,) = torch.autograd.grad(energy.sum(), pos_g, retain_graph=self.training)
        f = torch.empty_like(gpos)
        f[perm] = -gpos
        return (energy if self.training else energy.detach()), f
