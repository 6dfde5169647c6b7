"""Force head and parameter gradients (BASELINE.json configs[3]: QM9-style batch of 128 molecules, l_max = 2,
energy + force head).  -dE/dpos and dE/dparam from the HIP backward kernels (tensor products, edge geometry, gather,
gates, segment-sum) against torch autograd over the fp64 torch-CPU oracle (`segnn_oracle.energy_forces_torch`), plus
op-level checks of every new backward kernel against torch autograd of the same op written in torch.
Tolerance: fp32 kernels vs fp64 oracle, 2e-5 of the largest force / gradient magnitude (first derivatives accumulate the
forward's ~3e-7 rounding through ~10 chained products and two gates per layer)."""
import numpy as np
import pytest
import torch

import models  # noqa: F401
from oracle import graph_oracle as G
from oracle import segnn_oracle as S
from scalable_e3_gnn_amd import ops
from scalable_e3_gnn_amd.batched import BatchedEnergyModel, batched_radius_graph
from scalable_e3_gnn_amd.radius_graph import radius_graph

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _graph(N, k, seed):
    pos = torch.rand(N, 3, generator=torch.Generator().manual_seed(seed))
    r = float((3 * k / (4 * np.pi * N)) ** (1 / 3))
    return radius_graph(pos.to(DEV), r, [0, 0, 0], [1, 1, 1])


def _rel(a, b):
    a, b = a.detach(), b.detach()
    return float((a - b).abs().max() / b.abs().max())


@pytest.mark.parametrize("lmax", [1, 2])
def test_edge_geometry_backward_vs_torch(lmax):
    g = _graph(800, 10.0, 1)
    ny = (lmax + 1) ** 2
    pos = g.pos4[:, :3].clone().requires_grad_(True)
    Y, d, A = ops.edge_geometry(g, lmax=lmax, pos=pos)
    gen = torch.Generator(device=DEV).manual_seed(2)
    wY, wd, wA = (torch.randn(t.shape, device=DEV, generator=gen) for t in (Y, d, A))
    ((Y * wY).sum() + (d * wd).sum() + (A * wA).sum()).backward()
    got = pos.grad.clone()
    # the same three outputs in fp64 torch
    p64 = g.pos4[:, :3].double().clone().requires_grad_(True)
    src, dst = g.src.long(), g.dst.long()
    rel = p64[src] - p64[dst]
    dd = rel.norm(dim=1)
    u = rel / dd[:, None]
    parts = [torch.ones_like(dd)[:, None], 3 ** 0.5 * u]
    if lmax == 2:
        x, y, z = u[:, 0], u[:, 1], u[:, 2]
        s3 = 3 ** 0.5
        parts.append(5 ** 0.5 * torch.stack([s3 * x * y, s3 * y * z, (2 * z * z - x * x - y * y) / 2, s3 * z * x,
                                             s3 / 2 * (x * x - y * y)], 1))
    Y64 = torch.cat(parts, 1)
    deg = (g.rowptr[1:] - g.rowptr[:-1]).double().clamp_min(1)
    A64 = torch.cat([torch.ones(len(deg), 1, device=DEV, dtype=torch.float64),
                     torch.zeros(len(deg), ny - 1, device=DEV, dtype=torch.float64).index_add(0, dst, Y64[:, 1:]) / deg[:, None]], 1)
    assert _rel(Y.double(), Y64.detach()) < 1e-6 and _rel(A.double(), A64.detach()) < 1e-6
    ((Y64 * wY.double()).sum() + (dd * wd.double()).sum() + (A64 * wA.double()).sum()).backward()
    assert _rel(got.double(), p64.grad) < 2e-5


def test_gather_gate_segment_backward_vs_torch():
    g = _graph(600, 9.0, 3)
    E, N, D = g.num_edges, 600, 20
    gen = torch.Generator(device=DEV).manual_seed(4)
    h = torch.randn(N, D, device=DEV, generator=gen, requires_grad=True)
    dvec = torch.rand(E, device=DEV, generator=gen).requires_grad_(True)
    m = ops.gather_concat(h, g, dvec)                       # [E, 2D+1]
    t = ops.gate_blocks(m[:, :4 + 2 + 2 + 3 * 2 + 5 * 2].contiguous(), 4, [(1, 2), (2, 2)])   # 4 scalars, 2+2 gates, 1o x2, 2e x2
    t2 = ops.gate(m[:, 8:8 + 3 + 4 * 2].contiguous(), 3, 2)
    a = ops.segment_sum(torch.cat([t, t2], 1), g)
    w = torch.randn(a.shape, device=DEV, generator=gen)
    w3 = torch.randn(E, device=DEV, generator=gen)
    ((a * w).sum() + (m[:, -1] * w3).sum()).backward()           # the last column is the extra (distance) channel
    gh, gd = h.grad.clone(), dvec.grad.clone()
    # torch reference in fp64
    h64 = h.detach().double().requires_grad_(True)
    d64 = dvec.detach().double().requires_grad_(True)
    src, dst = g.src.long(), g.dst.long()
    m64 = torch.cat([h64[dst], h64[src], d64[:, None]], 1)

    def gb(x, ns, blocks):
        ng = sum(mm for _, mm in blocks)
        out, g0, c0 = [torch.nn.functional.silu(x[:, :ns])], ns, ns + ng
        for l, mm in blocks:
            wdt = 2 * l + 1
            out.append((torch.sigmoid(x[:, g0:g0 + mm])[:, :, None] * x[:, c0:c0 + mm * wdt].reshape(-1, mm, wdt)).reshape(-1, mm * wdt))
            g0, c0 = g0 + mm, c0 + mm * wdt
        return torch.cat(out, 1)

    t64 = gb(m64[:, :24], 4, [(1, 2), (2, 2)])
    t264 = gb(m64[:, 8:19], 3, [(1, 2)])
    a64 = torch.zeros(N, t64.shape[1] + t264.shape[1], device=DEV, dtype=torch.float64).index_add(0, dst, torch.cat([t64, t264], 1))
    assert _rel(a.detach().double(), a64.detach()) < 1e-6
    ((a64 * w.double()).sum() + (m64[:, -1] * w3.double()).sum()).backward()
    assert _rel(gh.double(), h64.grad) < 1e-5 and _rel(gd.double(), d64.grad) < 1e-5


def make_batch(seed, n_mol):
    rng = np.random.default_rng(seed)
    sizes = rng.integers(3, 30, n_mol)                       # QM9-shaped: 3..29 atoms
    pos = np.concatenate([rng.normal(size=(n, 3)) * 1.5 + rng.uniform(-40, 40, 3) for n in sizes]).astype(np.float32)
    batch = np.concatenate([np.full(n, i) for i, n in enumerate(sizes)])
    order = rng.permutation(len(batch))
    return pos[order], batch[order], sizes


@pytest.mark.parametrize("lmax,H,n_mol", [(2, 32, 128), (1, 16, 24)])
def test_energy_forces_and_parameter_grads_vs_fp64_oracle(lmax, H, n_mol):
    pos, batch, sizes = make_batch(5, n_mol)
    r, L = 5.0 if n_mol == 128 else 3.0, 2
    torch.manual_seed(6)
    model = BatchedEnergyModel("1x0e+1x1o", H, L, lmax=lmax).to(DEV)
    x = torch.randn(len(batch), 4, generator=torch.Generator().manual_seed(7))
    xd, pd, bd = x.to(DEV), torch.from_numpy(pos).to(DEV), torch.from_numpy(batch).to(DEV)
    model.eval()
    with torch.no_grad():                                    # forces switch grad on themselves
        e_fast = model(xd, pd, bd, r)                        # fused inference path
        e, f = model(xd, pd, bd, r, forces=True)
    assert f.shape == pd.shape
    # parameter gradients of the total energy
    model.train()
    model.zero_grad()
    model(xd, pd, bd, r).sum().backward()
    # oracle on the same graph (the lattice graph in the model's node order)
    g, mol = batched_radius_graph(pd, bd, r)
    perm = g.perm.cpu().numpy()
    params = {k[len("net."):]: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    E64, F64, gW = S.energy_forces_torch(params, H, L, lmax, "1x0e+1x1o", x.double().numpy()[perm],
                                         pos.astype(np.float64)[perm], g.rowptr.cpu().numpy(), g.src.cpu().numpy(),
                                         mol.cpu().numpy(), n_mol)
    F_want = np.empty_like(F64)
    F_want[perm] = F64
    scale_e = np.abs(E64).max()
    assert np.abs(e.detach().double().cpu().numpy() - E64).max() / scale_e < 1e-5
    assert np.abs(e_fast.double().cpu().numpy() - E64).max() / scale_e < 1e-5
    ferr = np.abs(f.double().cpu().numpy() - F_want).max() / np.abs(F_want).max()
    assert ferr < 2e-5, ferr
    worst = 0.0
    for name, p in model.named_parameters():
        want = gW[name[len("net."):]]
        assert p.grad is not None, name
        err = np.abs(p.grad.double().cpu().numpy() - want).max() / max(np.abs(want).max(), 1e-12)
        worst = max(worst, err)
        assert err < 2e-5, (name, err)
    print(f"\nconfigs[3]-style batch: {n_mol} molecules, {len(batch)} atoms, E={g.num_edges} edges, l_max={lmax} H={H}: "
          f"force err {ferr:.2e}, worst parameter-gradient err {worst:.2e} (vs fp64 torch-CPU autograd oracle)")
    # translation invariance of the energy <=> forces of every molecule sum to zero
    fsum = torch.zeros(n_mol, 3, device=DEV).index_add_(0, bd.long(), f)
    assert float(fsum.abs().max()) < 1e-3 * float(f.abs().max())
