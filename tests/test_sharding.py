"""N > 1 path on CPU: world-size-2 gloo.  The strips come from the oracle (the
product has no CPU path); what is under test is the sharding logic that bench.py
and a multi-GPU caller use: sector bounds, the gather, the reassembly."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from horizonator_amd.sharding import (StripExchange, agree_on_capacity, azimuth_density, balanced_layout, broadcast_dem, gather_flat_async, gather_strips, gather_strips_async, gather_viewpoints,
                                      gatherer_weights, sector_columns, sparse_header_words, sparse_mask_stride,
                                      viewpoint_slice)

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("width,world", [(16000, 8), (1003, 8), (7, 2), (5, 5), (2000, 3)])
def test_sectors_partition_the_columns(width, world):
    cols = [sector_columns(width, world, r) for r in range(world)]
    assert cols[0][0] == 0 and cols[-1][1] == width
    for a, b in zip(cols, cols[1:]):
        assert a[1] == b[0]
    sizes = [c1 - c0 for c0, c1 in cols]
    assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sector_columns(width, world, world)


def test_weighted_sectors_partition_the_columns():
    """a gathering rank that also converts the panorama draws less - or nothing"""
    for width, weights in [(16000, [0.5, 1, 1, 1]), (1003, [0, 1, 1]), (7, [1, 0, 3]), (100, [2.5])]:
        world = len(weights)
        cols = [sector_columns(width, world, r, weights) for r in range(world)]
        assert cols[0][0] == 0 and cols[-1][1] == width
        assert all(a[1] == b[0] for a, b in zip(cols, cols[1:]))
        assert all(c1 >= c0 for c0, c1 in cols)
        for (c0, c1), w in zip(cols, weights):
            assert abs((c1 - c0) - width * w / sum(weights)) <= 1
    assert sector_columns(1003, 3, 0, [0, 1, 1]) == (0, 0)
    with pytest.raises(ValueError):
        sector_columns(10, 2, 0, [0, 0])
    with pytest.raises(ValueError):
        sector_columns(10, 2, 0, [1, -1])
    # rank 0's share shrinks with the number of ranks and reaches zero
    shares = [gatherer_weights(g, 1.77, 0.27)[0] for g in (2, 4, 8)]
    assert 1 > shares[0] > shares[1] > shares[2] == 0.0
    assert gatherer_weights(1, 1.77, 0.27) == [1.0]
    assert gatherer_weights(4, 1.0, 0.0) == [1.0] * 4


def test_balanced_layouts_deal_equal_work():
    """sectors by work instead of by azimuth: the geometric density of a square DEM window, and
    arbitrary measured densities"""
    W = 16000
    d = azimuth_density(W, -180.0, 180.0, np.cos(np.radians(34.4)), floor=0.1)
    assert d.shape == (W,) and d.min() > 0
    lay = balanced_layout(d, 8)
    assert lay[0][0] == 0 and lay[-1][1] == W and all(a[1] == b[0] for a, b in zip(lay, lay[1:]))
    shares = [d[c0:c1].sum() / d.sum() for c0, c1 in lay]
    assert max(shares) - min(shares) < 2e-3
    widths = [c1 - c0 for c0, c1 in lay]
    # north and south (image edges and centre) hold more terrain per degree than east and west
    assert widths[0] < widths[1] and widths[3] < widths[2] and widths[4] < widths[5] and widths[7] < widths[6]
    # by symmetry 2 and 4 ranks get equal sectors
    assert all(abs((c1 - c0) - W // 4) <= 2 for c0, c1 in balanced_layout(d, 4))
    # rank weights scale the shares; weight 0 = no columns
    lay = balanced_layout(np.ones(1000), 4, [0.0, 1.0, 1.0, 2.0])
    assert lay == [(0, 0), (0, 250), (250, 500), (500, 1000)]
    # a step density: twice the cost on the left half
    lay = balanced_layout(np.concatenate([np.full(500, 2.0), np.full(500, 1.0)]), 3)
    assert lay == [(0, 250), (250, 500), (500, 1000)]
    with pytest.raises(ValueError):
        balanced_layout(np.zeros(10), 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = np.load(os.path.join(GOLD, "render_G4_move2.npz"))
        W, H = int(g["W"]) - 1, int(g["H"])          # odd width: strips differ by one column
        v = oracle.make_view(**{k: float(g["u_" + k]) for k in oracle.VIEW_FIELDS})
        c0, c1 = sector_columns(W, world, rank)
        mine = oracle.render(g["mosaic"], v, W, H, c0, c1, nthreads=1)
        img = gather_strips(torch.from_numpy(mine["bgr"]), W)
        rng = gather_strips(torch.from_numpy(mine["ranges"]), W)
        idx = gather_strips(torch.from_numpy(mine["index"]), W)
        # the pipelined form bench.py uses: two exchanges in flight, finished later
        h1 = gather_strips_async(torch.from_numpy(mine["z24"].astype(np.int64)), W)
        h2 = gather_strips_async(torch.from_numpy(mine["ranges"]), W)
        z_async, rng_async = h1.result(), h2.result()
        # strips as one packed word per pixel (z24<<8 | red8), read in place on rank 0: what bench.py gathers
        packed = (mine["z24"].astype(np.int64) << 8) | mine["bgr"][..., 2].astype(np.int64)
        parts = gather_strips_async(torch.from_numpy(packed), W).parts()
        if rank == 0:
            whole = np.zeros((H, W), np.int64)
            for t, c0, n in parts:
                whole[:, c0:c0 + n] = t[:, :n].numpy()
            ref = oracle.render(g["mosaic"], v, W, H, nthreads=1)
            assert np.array_equal(whole, (ref["z24"].astype(np.int64) << 8) | ref["bgr"][..., 2].astype(np.int64))
        else:
            assert parts is None
        # DEM distribution: rank 0 has the window and the mosaic, everybody ends up with both
        win0 = (1200, g["mosaic"].shape[0] // 2, -118, 34, 7, 11)
        bw, bm = broadcast_dem(win0 if rank == 0 else None, g["mosaic"] if rank == 0 else None)
        assert bw == win0 and bm.dtype == np.int16 and np.array_equal(bm, g["mosaic"])
        # 1-D buffers of one agreed length (the sparse strips of bench.py)
        mine_len = torch.tensor([100 + 50 * rank], dtype=torch.int64)
        dist.all_reduce(mine_len, op=dist.ReduceOp.MAX)
        flat = torch.arange(int(mine_len), dtype=torch.int32) + 1000 * rank
        got = gather_flat_async(flat).tensors()
        if rank == 0:
            assert len(got) == world and all(np.array_equal(t.numpy(), np.arange(150, dtype=np.int32) + 1000 * r)
                                             for r, t in enumerate(got))
        else:
            assert got is None
        assert sparse_mask_stride(33) == 2 and sparse_header_words(10, 2) == 31
        # the same without a host round trip per panorama: one agreed capacity, preallocated bins, the
        # strip's own first word says how much of it counts; a strip that does not fit is flagged
        # beside the data and the exchange redone with more room
        HDR, FULL = 8, 4000

        def strip_of(r, t, salt):
            b = torch.zeros(FULL, dtype=torch.int32)
            b[0] = t
            b[1:HDR] = 7 * r + salt
            b[HDR:HDR + t] = torch.arange(t, dtype=torch.int32) + 100000 * r + salt
            return b
        cap = agree_on_capacity(HDR + 300 + 100 * rank, HDR, FULL, torch.device("cpu"))
        assert cap == int(HDR + 400 * 1.1 + 1024) and cap < FULL
        ex = StripExchange(cap, FULL, HDR, torch.device("cpu"), nslots=2)
        sent = [strip_of(rank, 300 + 100 * rank, 1), strip_of(rank, 250 + 30 * rank, 2)]
        ex.post(0, sent[0])
        ex.post(1, sent[1])                                 # two panoramas in flight
        for slot, salt, counts in ((0, 1, (300, 400)), (1, 2, (250, 280))):
            bins, overflow = ex.complete(slot)
            assert not overflow
            if rank == 0:
                for r, b in enumerate(bins):
                    assert b.numel() == cap and int(b[0]) == counts[r]
                    assert np.array_equal(b[:HDR + counts[r]].numpy(), strip_of(r, counts[r], salt)[:HDR + counts[r]].numpy())
            else:
                assert bins is None
        # rank 1 suddenly sees much more terrain than the capacity allows for
        big = strip_of(rank, 3000 if rank == 1 else 200, 3)
        ex.post(0, big)
        bins, overflow = ex.complete(0)
        assert overflow                                     # ... on every rank
        bins = ex.grow(0, big)
        assert ex.resends == 1 and ex.cap >= HDR + 3000
        if rank == 0:
            for r, t in enumerate((200, 3000)):
                assert np.array_equal(bins[r][:HDR + t].numpy(), strip_of(r, t, 3)[:HDR + t].numpy())
        # ... and the way bench.py runs it: the OTHER slot's exchange is still in flight (posted, completed a
        # panorama later) when a strip outgrows the capacity.  grow() reallocates its own slot only; the
        # exchange in flight keeps its bins and arrives intact, and its slot adopts the capacity when posted next.
        ex2 = StripExchange(HDR + 600, FULL, HDR, torch.device("cpu"), nslots=2)
        inflight = strip_of(rank, 500, 4)
        ex2.post(1, inflight)
        huge = strip_of(rank, 3500 if rank == 0 else 100, 5)
        ex2.post(0, huge)
        bins0, overflow = ex2.complete(0)
        assert overflow
        old_cap = ex2.cap
        bins0 = ex2.grow(0, huge)                           # slot 1 is posted and not completed
        assert ex2.cap > old_cap and ex2.slot_cap == [ex2.cap, old_cap]
        bins1, overflow1 = ex2.complete(1)
        assert not overflow1
        if rank == 0:
            for r, t in enumerate((3500, 100)):
                assert np.array_equal(bins0[r][:HDR + t].numpy(), strip_of(r, t, 5)[:HDR + t].numpy())
            for r in range(world):
                assert bins1[r].numel() == old_cap
                assert np.array_equal(bins1[r][:HDR + 500].numpy(), strip_of(r, 500, 4)[:HDR + 500].numpy())
        ex2.post(1, huge)                                   # the slot takes the new capacity over
        bins1, overflow1 = ex2.complete(1)
        assert not overflow1 and ex2.slot_cap == [ex2.cap, ex2.cap]
        if rank == 0:
            assert np.array_equal(bins1[0][:HDR + 3500].numpy(), strip_of(0, 3500, 5)[:HDR + 3500].numpy())
        # the gathering rank rotates from panorama to panorama (bench.py --gather rotate): every rank keeps bins
        ex3 = StripExchange(HDR + 600, FULL, HDR, torch.device("cpu"), nslots=2, any_dst=True)
        for k in range(4):
            slot, dst = k % 2, k % world
            ex3.post(slot, strip_of(rank, 200 + 10 * rank + k, 6 + k), dst=dst)
            binsk, overflowk = ex3.complete(slot)
            assert not overflowk
            if rank == dst:
                for r in range(world):
                    t = 200 + 10 * r + k
                    assert np.array_equal(binsk[r][:HDR + t].numpy(), strip_of(r, t, 6 + k)[:HDR + t].numpy())
            else:
                assert binsk is None
        # unequal sectors, rank 0 drawing nothing at all
        wts = [0.0, 1.0]
        d0, d1 = sector_columns(W, world, rank, wts)
        part = oracle.render(g["mosaic"], v, W, H, d0, d1, nthreads=1, want=("z24",))["z24"].astype(np.int64) \
            if d1 > d0 else np.zeros((H, 0), np.int64)
        wparts = gather_strips_async(torch.from_numpy(part), W, weights=wts).parts()
        if rank == 0:
            assert [(c0, n) for _, c0, n in wparts] == [(0, 0), (0, W)]
            assert np.array_equal(wparts[1][0][:, :W].numpy(), oracle.render(g["mosaic"], v, W, H, nthreads=1)["z24"].astype(np.int64))
        if rank == 0:
            full = oracle.render(g["mosaic"], v, W, H, nthreads=1)
            assert np.array_equal(z_async.numpy(), full["z24"].astype(np.int64))
            assert np.array_equal(rng_async.numpy(), full["ranges"])
            ok = (np.array_equal(img.numpy(), full["bgr"]) and np.array_equal(rng.numpy(), full["ranges"])
                  and np.array_equal(idx.numpy(), full["index"]))
            q.put(bool(ok))
        else:
            assert img is None and rng is None and z_async is None and rng_async is None
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_gather_reassembles_the_panorama():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert q.get(timeout=5) is True


# ---- batch of viewpoints, sharded by viewpoint (BASELINE.json configs[3]) -------

def _batch_views(g, n):
    """n viewpoints around the fixture's: the viewer cell moves, everything else stays"""
    import oracle
    base = {k: float(g["u_" + k]) for k in oracle.VIEW_FIELDS}
    views = []
    for v in range(n):
        u = dict(base)
        u["viewer_cell_i"] = base["viewer_cell_i"] + 1.75 * (v % 3 - 1)
        u["viewer_cell_j"] = base["viewer_cell_j"] - 1.25 * (v // 3 - 1)
        views.append(oracle.make_view(**u))
    return views


def _batch_worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = np.load(os.path.join(GOLD, "render_G4_move2.npz"))
        W, H, n = int(g["W"]), int(g["H"]), 7            # 7 over 2 ranks: blocks of 4 and 3
        views = _batch_views(g, n)
        v0, v1 = viewpoint_slice(n, world, rank)
        mine = np.stack([oracle.render(g["mosaic"], views[v], W, H, nthreads=1, want=("bgr",))["bgr"]
                         for v in range(v0, v1)])
        batch = gather_viewpoints(torch.from_numpy(mine), n)
        # ... in chunks smaller than a rank's block (3 of 4), and into a result the caller owns
        mine_t = torch.from_numpy(mine)
        own = torch.zeros((n,) + tuple(mine_t.shape[1:]), dtype=mine_t.dtype) if rank == 0 else None
        chunked = gather_viewpoints(mine_t, n, chunk=3, out=own)
        one_by_one = gather_viewpoints(mine_t, n, chunk=1)
        if rank == 0:
            assert chunked is own and torch.equal(chunked, batch) and torch.equal(one_by_one, batch)
        else:
            assert chunked is None and one_by_one is None
        if rank == 0:
            assert batch.shape == (n, H, W, 3)
            ok = all(np.array_equal(batch[v].numpy(),
                                    oracle.render(g["mosaic"], views[v], W, H, nthreads=1, want=("bgr",))["bgr"])
                     for v in range(n))
            distinct = len({batch[v].numpy().tobytes() for v in range(n)}) == n
            q.put(bool(ok and distinct))
        else:
            assert batch is None
    finally:
        dist.destroy_process_group()


def test_viewpoint_slices_partition_the_batch():
    for n, world in [(256, 8), (256, 3), (7, 2), (3, 8)]:
        sl = [viewpoint_slice(n, world, r) for r in range(world)]
        assert sl[0][0] == 0 and sl[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(sl, sl[1:]))


def test_two_rank_gloo_gather_of_a_viewpoint_batch():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_batch_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert q.get(timeout=5) is True
