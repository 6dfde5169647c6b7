"""gloo tests (world_size 2 and 4, CPU) of the spatial sharding: slab partition + ghost halo + split edge lists, product
code in scalable-e3-gnn_amd/sharding.py.  The GPU kernels cannot run here, so the per-rank compute uses the numpy
oracle; what is under test is the partition, the ghost bookkeeping across the Morton renumbering (ranks with ONE and
with TWO neighbours, ranks whose halo is empty), the in-place per-layer refresh and the interior / boundary split: the
sharded forward on the split graph must equal the unsharded oracle forward."""
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


def _cloud(kind, N, world):
    g = torch.Generator().manual_seed(7)
    pos = torch.rand(N, 3, generator=g, dtype=torch.float64)
    if kind == "uniform":
        pos[:, 0] *= world                                   # global box [0,world) x [0,1)^2
    elif kind == "clustered":
        # non-uniform: 70 % of the particles in a blob that straddles the 1|2 face, rank 3's slab almost empty
        blob = torch.rand(N, generator=g) < 0.7
        pos[:, 0] = torch.where(blob, 2.0 + 0.35 * torch.randn(N, generator=g, dtype=torch.float64), pos[:, 0] * world)
        pos[:, 0].clamp_(0.0, world - 1e-9)
    elif kind == "gap":
        # rank 1's slab holds nothing near its faces and rank 2's slab is EMPTY: empty halos and empty messages
        u = torch.rand(N, generator=g, dtype=torch.float64)
        pos[:, 0] = torch.where(u < 0.5, 0.9 * u / 0.5, torch.where(u < 0.6, 1.4 + 0.2 * (u - 0.5) / 0.1, 3.0 + (u - 0.6) / 0.4))
    x = torch.randn(N, 4, generator=g, dtype=torch.float64)
    return pos, x


def _worker(rank, world, port, N, H, L, kind, out_q):
    sys.path.insert(0, REPO)
    import models  # noqa
    from oracle import graph_oracle as G
    from oracle import segnn_oracle as S
    from scalable_e3_gnn_amd.radius_graph import RadiusGraph
    from scalable_e3_gnn_amd.segnn import SEGNN
    from scalable_e3_gnn_amd.sharding import SlabHalo

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pos, x = _cloud(kind, N, world)
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
        nloc = lpos.shape[0]
        if nloc:
            perm, rowptr, src = G.graph(lpos.numpy(), lo, hi, r)
        else:
            perm, rowptr, src = np.zeros(0, np.int32), np.zeros(1, np.int32), np.zeros(0, np.int32)
        halo.renumber(torch.as_tensor(perm))
        lp, lxx = lpos.numpy().astype(np.float32)[perm], lx.numpy()[perm]
        # split graph: edges into ghost rows dropped, the rest = interior (owned src) + boundary (ghost src)
        g = RadiusGraph(torch.as_tensor(perm), torch.zeros(nloc, 4), torch.as_tensor(rowptr), torch.as_tensor(src),
                        len(src), ((1, 1, 1), 0))
        sp = halo.split_graph(g)
        gs, gd = sp.graph.src.numpy(), sp.graph.dst.numpy()
        ghost = halo.is_ghost.numpy()
        assert not ghost[gd].any() and sp.dropped == len(src) - len(gs)
        assert np.array_equal(np.diff(sp.graph.rowptr.numpy()), np.bincount(gd, minlength=nloc))
        (isrc, idst), (bsrc, bdst) = [(a.numpy(), b.numpy()) for a, b in (sp.interior, sp.boundary)]
        assert not ghost[isrc].any() and (len(bsrc) == 0 or ghost[bsrc].all())
        assert len(isrc) + len(bsrc) == len(gs)
        both = np.concatenate([np.stack([idst, isrc], 1), np.stack([bdst, bsrc], 1)])
        assert np.array_equal(both[np.lexsort((both[:, 1], both[:, 0]))], np.stack([gd, gs], 1))  # same edge multiset
        assert np.all(np.diff(idst) >= 0) and np.all(np.diff(bdst) >= 0)                          # both still dst-sorted

        calls = []

        def exchange(h):
            t = torch.as_tensor(h)
            before = t.data_ptr()
            tok = halo.start(t)                              # the overlapped form: post, (compute), finish in place
            out = halo.finish(t, tok)
            assert out.data_ptr() == before                  # refreshed in place, no clone of h
            calls.append(1)
            return out.numpy()

        if nloc:
            out = S.forward(params, H, L, "1x0e+1x1o", "1x1o", lxx, lp, sp.graph.rowptr.numpy(), gs, exchange=exchange)
            owned_out = out[halo.owned_new.numpy()]          # back to the owned particles' original order
        else:
            for _ in range(L):
                exchange(np.zeros((0, 4 * H)))               # an empty rank still takes part in every exchange
            owned_out = np.zeros((0, 3))
        assert len(calls) == L
        if rank == 0:
            # unsharded reference on the whole cloud
            gperm, grp, gsrc = G.graph(pos.numpy(), [0, 0, 0], [world, 1, 1], r)
            want = S.forward(params, H, L, "1x0e+1x1o", "1x1o", x.numpy()[gperm], pos.numpy().astype(np.float32)[gperm], grp, gsrc)
            full = np.empty_like(want)
            full[gperm] = want                               # original particle order
            out_q.put(("ref", full, None))
        out_q.put(("part", owned_out, own.numpy()))
        out_q.put(("halo", np.array([halo.n_ghost_left, halo.n_ghost_right, halo.bytes_last_exchange, own.numel(),
                                     sp.dropped, len(isrc), len(bsrc)]), rank))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _run(world, N, H, L, kind):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, H, L, kind, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=280) for _ in range(1 + 2 * world)]
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
    return {idx: val for tag, val, idx in got if tag == "halo"}


@pytest.mark.timeout(300)
def test_sharded_forward_equals_unsharded_world2():
    halos = _run(2, 1600, 4, 3, "uniform")
    assert halos[0][0] == 0 and halos[1][1] == 0            # outer faces have no ghosts
    assert halos[0][1] > 0 and halos[1][0] > 0 and halos[0][2] > 0
    assert all(h[4] > 0 and h[5] > 0 and h[6] > 0 for h in halos.values())  # dropped / interior / boundary edges all occur


@pytest.mark.timeout(300)
def test_sharded_forward_world4_middle_ranks_have_two_neighbours():
    halos = _run(4, 2400, 4, 2, "uniform")
    assert halos[0][0] == 0 and halos[3][1] == 0
    for k in (1, 2):                                         # middle ranks: ghosts from BOTH sides
        assert halos[k][0] > 0 and halos[k][1] > 0


@pytest.mark.timeout(300)
def test_sharded_forward_world4_clustered_cloud():
    halos = _run(4, 2400, 4, 2, "clustered")
    own = [int(halos[k][3]) for k in range(4)]
    assert max(own) > 3 * max(1, min(own))                   # the partition really is unbalanced


@pytest.mark.timeout(300)
def test_sharded_forward_world4_empty_rank_and_empty_halos():
    halos = _run(4, 2000, 4, 2, "gap")
    assert halos[2][3] == 0                                  # rank 2 owns nothing
    assert halos[1][0] == 0 or halos[1][1] == 0              # rank 1 has an empty halo on at least one side
    assert halos[3][0] == 0                                  # nothing arrives from the empty slab


# ---------------------------------------------------------------------------------------------------------------------
# ONE cloud cut into 2 x 2 x 2 octants (= the top level of the Morton order): the strong-scaling layout; neighbour sets > 2
# ---------------------------------------------------------------------------------------------------------------------
def _octant_worker(rank, world, port, N, H, L, out_q):
    sys.path.insert(0, REPO)
    import models  # noqa
    from oracle import graph_oracle as G
    from oracle import segnn_oracle as S
    from scalable_e3_gnn_amd.radius_graph import RadiusGraph
    from scalable_e3_gnn_amd.segnn import SEGNN
    from scalable_e3_gnn_amd.sharding import GridHalo

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g0 = torch.Generator().manual_seed(11)
        pos = torch.rand(N, 3, generator=g0, dtype=torch.float64)
        x = torch.randn(N, 4, generator=g0, dtype=torch.float64)
        r = float((3 * 10.0 / (4 * np.pi * N)) ** (1 / 3))
        torch.manual_seed(0)
        model = SEGNN("1x0e+1x1o", H, "1x1o", L)
        params = {k: v.detach().double().numpy() for k, v in model.state_dict().items()}

        halo = GridHalo((2, 2, 2), (0, 0, 0), (1, 1, 1))
        assert len(halo.neighbours) == 7                      # every octant touches the 7 others
        own = (halo.owner_of(pos) == rank).nonzero().flatten()
        lpos, lx = halo.setup(pos[own].float().double(), x[own], r)
        blo, bhi = halo.box(rank)
        lo, hi = [v - 2 * r for v in blo], [v + 2 * r for v in bhi]
        nloc = lpos.shape[0]
        perm, rowptr, src = G.graph(lpos.numpy(), lo, hi, r)
        halo.renumber(torch.as_tensor(perm))
        lp, lxx = lpos.numpy().astype(np.float32)[perm], lx.numpy()[perm]
        g = RadiusGraph(torch.as_tensor(perm), torch.zeros(nloc, 4), torch.as_tensor(rowptr), torch.as_tensor(src),
                        len(src), ((1, 1, 1), 0))
        sp = halo.split_graph(g)
        ghost = halo.is_ghost.numpy()
        assert not ghost[sp.graph.dst.numpy()].any()
        assert len(sp.interior[0]) + len(sp.boundary[0]) == sp.graph.num_edges

        def exchange(h):
            t = torch.as_tensor(h)
            return halo.finish(t, halo.start(t)).numpy()

        out = S.forward(params, H, L, "1x0e+1x1o", "1x1o", lxx, lp, sp.graph.rowptr.numpy(), sp.graph.src.numpy(),
                        exchange=exchange)
        owned_out = out[halo.owned_new.numpy()]
        # the same cloud cut into 8 slabs along x, for the ghost-fraction comparison only (no forward)
        slab = GridHalo((8, 1, 1), (0, -1e30, -1e30), (1, 1e30, 1e30))
        sown = (slab.owner_of(pos) == rank).nonzero().flatten()
        slab.setup(pos[sown].float().double(), x[sown], r)
        if rank == 0:
            gperm, grp, gsrc = G.graph(pos.numpy(), [0, 0, 0], [1, 1, 1], r)
            want = S.forward(params, H, L, "1x0e+1x1o", "1x1o", x.numpy()[gperm], pos.numpy().astype(np.float32)[gperm], grp, gsrc)
            full = np.empty_like(want)
            full[gperm] = want
            out_q.put(("ref", full, None))
        out_q.put(("part", owned_out, own.numpy()))
        out_q.put(("halo", np.array([halo.n_ghost, own.numel(), sum(1 for c in halo.recv_counts if c > 0), slab.n_ghost,
                                     sown.numel(), sum(1 for c in slab.recv_counts if c > 0), r]), rank))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(420)
def test_octant_partition_world8_equals_unsharded():
    """8 ranks, ONE unit cube: sharded forward on octants == unsharded forward; ghost fraction of octants vs 8 x-slabs."""
    world, N, H, L = 8, 6000, 4, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_octant_worker, args=(r, world, port, N, H, L, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=400) for _ in range(1 + 2 * world)]
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
    halos = np.stack([val for tag, val, idx in got if tag == "halo"])
    oct_frac, slab_frac = halos[:, 0].sum() / halos[:, 1].sum(), halos[:, 3].sum() / halos[:, 4].sum()
    r = halos[0, 6]
    print(f"\nworld 8, one unit cube, N={N}, r={r:.3f}: octants ghosts/owned = {oct_frac:.3f} over {halos[:, 2].mean():.1f} "
          f"neighbours per rank; 8 x-slabs ghosts/owned = {slab_frac:.3f} over {halos[:, 5].mean():.1f} neighbours per rank")
    assert (halos[:, 2] >= 3).all()                           # neighbour sets > 2: faces, edges and the corner
    assert oct_frac < slab_frac                               # fewer ghosts than slabs of width 1/8, spread over more links
