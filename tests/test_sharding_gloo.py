"""world_size-2 gloo test of the spatial sharding (slab partition + ghost halo, product code in
scalable-e3-gnn_amd/sharding.py).  The GPU kernels cannot run here, so the per-rank compute uses the
numpy oracle; what is under test is the partition, the ghost bookkeeping across the Morton renumbering
and the per-layer exchange: the sharded forward must equal the unsharded oracle forward."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, N, H, L, out_q):
    sys.path.insert(0, REPO)
    import models  # noqa
    from oracle import graph_oracle as G
    from oracle import segnn_oracle as S
    from scalable_e3_gnn_amd.segnn import SEGNN
    from scalable_e3_gnn_amd.sharding import SlabHalo

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(7)
        pos = torch.rand(N, 3, generator=g, dtype=torch.float64)
        pos[:, 0] *= world                                   # global box [0,world) x [0,1)^2
        x = torch.randn(N, 4, generator=g, dtype=torch.float64)
        r = float((3 * 10.0 / (4 * np.pi * (N / world))) ** (1 / 3))
        torch.manual_seed(0)
        model = SEGNN("1x0e+1x1o", H, "1x1o", L)             # ctor only (no GPU needed)
        params = {k: v.detach().double().numpy() for k, v in model.state_dict().items()}

        own = ((pos[:, 0] >= rank) & (pos[:, 0] < rank + 1)).nonzero().flatten()
        halo = SlabHalo()
        lpos, lx = halo.setup(pos[own].float().double(), x[own], float(rank), float(rank + 1), r)
        # features of another storage type than the positions keep their dtype through the exchange (bf16 storage
        # with fp32 positions is what `bench.py --gpus N` sends for its bf16 leg)
        h2 = SlabHalo()
        p32, f16 = h2.setup(pos[own].float(), x[own].to(torch.bfloat16), float(rank), float(rank + 1), r)
        assert p32.dtype == torch.float32 and f16.dtype == torch.bfloat16 and f16.shape[0] == p32.shape[0]
        assert torch.equal(f16[: own.numel()], x[own].to(torch.bfloat16))
        lo, hi = [rank - 2 * r, 0, 0], [rank + 1 + 2 * r, 1, 1]
        perm, rowptr, src = G.graph(lpos.numpy(), lo, hi, r)
        halo.renumber(torch.as_tensor(perm))
        lp, lxx = lpos.numpy().astype(np.float32)[perm], lx.numpy()[perm]

        def exchange(h):
            return halo.exchange(torch.as_tensor(h)).numpy()

        out = S.forward(params, H, L, "1x0e+1x1o", "1x1o", lxx, lp, rowptr, src, exchange=exchange)
        owned_out = out[halo.owned_new.numpy()]              # back to the owned particles' original order
        if rank == 0:
            # unsharded reference on the whole cloud
            gperm, grp, gsrc = G.graph(pos.numpy(), [0, 0, 0], [world, 1, 1], r)
            want = S.forward(params, H, L, "1x0e+1x1o", "1x1o", x.numpy()[gperm], pos.numpy().astype(np.float32)[gperm], grp, gsrc)
            full = np.empty_like(want)
            full[gperm] = want                               # original particle order
            out_q.put(("ref", full, None))
        out_q.put(("part", owned_out, own.numpy()))
        out_q.put(("halo", np.array([halo.n_ghost_left, halo.n_ghost_right, halo.bytes_last_exchange]), rank))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_forward_equals_unsharded_world2():
    world, N, H, L = 2, 1600, 4, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, H, L, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in range(1 + 2 * world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = [g for g in got if g[0] == "ref"][0][1]
    merged = np.full_like(ref, np.nan)
    for tag, val, idx in got:
        if tag == "part":
            merged[idx] = val
    assert not np.isnan(merged).any(), "every particle must be owned by exactly one rank"
    assert np.abs(merged - ref).max() / np.abs(ref).max() < 1e-10
    halos = {idx: val for tag, val, idx in got if tag == "halo"}
    assert halos[0][0] == 0 and halos[1][1] == 0            # outer faces have no ghosts
    assert halos[0][1] > 0 and halos[1][0] > 0 and halos[0][2] > 0
