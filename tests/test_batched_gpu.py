"""Batched small-graph path (configs[3]): one lattice radius graph for a batch of molecules + energy head, against
per-molecule runs of the fp64 oracle (numpy brute-force graph + oracle SEGNN)."""
import numpy as np
import pytest
import torch

import models  # noqa: F401
from oracle import graph_oracle as G
from oracle import segnn_oracle as S
from scalable_e3_gnn_amd.batched import BatchedEnergyModel, batched_radius_graph

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def make_batch(seed, n_mol=9):
    rng = np.random.default_rng(seed)
    sizes = rng.integers(3, 14, n_mol)
    pos = np.concatenate([rng.uniform(0, 2.5, (n, 3)) + rng.uniform(-50, 50, 3) for n in sizes]).astype(np.float32)
    batch = np.concatenate([np.full(n, i) for i, n in enumerate(sizes)])
    order = rng.permutation(len(batch))          # atoms of a molecule need not be contiguous
    return pos[order], batch[order], sizes


def test_batched_graph_has_exactly_the_intramolecular_pairs():
    pos, batch, sizes = make_batch(1)
    r = 1.4
    g, mol = batched_radius_graph(torch.from_numpy(pos).to(DEV), torch.from_numpy(batch).to(DEV), r)
    perm = g.perm.cpu().numpy()
    assert np.array_equal(mol.cpu().numpy(), batch[perm])
    assert np.array_equal(g.pos4[:, :3].cpu().numpy(), pos[perm])          # original coordinates, exact
    rowptr, src = g.rowptr.cpu().numpy(), g.src.cpu().numpy()
    got = set()
    for i in range(len(perm)):
        for s in src[rowptr[i]:rowptr[i + 1]]:
            got.add((int(perm[i]), int(perm[s])))
    want = set()
    for m in range(len(sizes)):
        ids = np.nonzero(batch == m)[0]
        rp, sc = G.graph_numpy(pos[ids], r)
        for a in range(len(ids)):
            for b in sc[rp[a]:rp[a + 1]]:
                want.add((int(ids[a]), int(ids[b])))
    assert got == want and len(want) > 0


@pytest.mark.parametrize("H", [8, 32])
def test_batched_energy_vs_per_molecule_oracle(H):
    pos, batch, sizes = make_batch(2)
    r, L = 1.4, 2
    torch.manual_seed(3)
    model = BatchedEnergyModel("1x0e+1x1o", H, L, lmax=2).to(DEV)
    x = torch.randn(len(batch), 4, generator=torch.Generator().manual_seed(4))
    with torch.no_grad():
        e = model(x.to(DEV), torch.from_numpy(pos).to(DEV), torch.from_numpy(batch).to(DEV), r).double().cpu().numpy()
    params = {k[len("net."):]: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    want = np.zeros(len(sizes))
    for m in range(len(sizes)):
        ids = np.nonzero(batch == m)[0]
        rp, sc = G.graph_numpy(pos[ids], r)
        out = S.forward_l2(params, H, L, "1x0e+1x1o", "1x0e", x.double().numpy()[ids], pos[ids].astype(np.float64), rp, sc)
        want[m] = out[:, 0].sum()
    assert np.abs(e - want).max() / np.abs(want).max() < 1e-5
