"""world_size-2 gloo tests of the multi-GPU host logic (runs on CPU).  The per-rank compute here is the CPU oracle --
the point is the partition / gather / all-reduce plumbing, which is backend-independent."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gens_amd import distributed as D

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ray_shards_partition_the_rays():
    for n in (0, 1, 7, 24, 307200):
        for world in (1, 2, 3, 8):
            spans = [D.ray_shard(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(e - s for s, e in spans) - min(e - s for s, e in spans) <= 1
    assert D.scene_shard(15, 1, 8) == [1, 9]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, tag, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import render_oracle as R
        d = np.load(os.path.join(ROOT, "tests", "golden", tag + ".npz"))
        g = {k: torch.from_numpy(d[k]) for k in d.files}
        sd = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
        feats = [g[f"feat{i}"] for i in range(5)]
        vols = [g[f"vol{i}"] for i in range(3)]
        masks = [g[f"mask{i}"] for i in range(3)]
        n = 10
        ro, rd, jit = g["rays_o"][:n], g["rays_d"][:n], g["draw_trand"][:n]
        pts_rand = g["draw_ptsrand"] * 2 - 1

        def render(o, dr, j):
            out = R.render(sd, o, dr, g["near"], g["far"], vols, masks, g["imgs"], feats, feats, g["intrs"], g["c2ws"], 1.0, None, j, pts_rand)
            return {"color": out["color_fine"].detach(), "depth": out["render_depth"].detach()[:, None]}
        full = D.render_sharded(render, ro, rd, jit)
        # gradient all-reduce: each rank holds rank-dependent grads, result must be their mean
        p = torch.nn.Parameter(torch.zeros(5))
        p.grad = torch.full((5,), float(rank + 1))
        q_ = torch.nn.Parameter(torch.zeros(2, 3))
        q_.grad = torch.full((2, 3), float(10 * (rank + 1)))
        D.allreduce_gradients([p, q_])
        if rank == 0:
            ref = render(ro, rd, jit)
            q.put(tuple(t.numpy().copy() for t in (full["color"], full["depth"], ref["color"], ref["depth"], p.grad, q_.grad)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_sharded_render_equals_single_process_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, "g9b_render", q)) for r in range(world)]
    for p in procs:
        p.start()
    color, depth, ref_color, ref_depth, g1, g2 = (torch.from_numpy(a) for a in q.get(timeout=500))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # rays are independent, so the partition cannot change the image (only float round-off of batch-shaped GEMMs)
    assert (color - ref_color).abs().max() < 1e-5
    assert (depth - ref_depth).abs().max() < 1e-5
    assert torch.allclose(g1, torch.full((5,), 1.5)) and torch.allclose(g2, torch.full((2, 3), 15.0))


def _worker_flat(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        vols = [torch.nn.Parameter(torch.zeros(1, 4, d, d, d)) for d in (6, 3)]
        w = torch.nn.Parameter(torch.zeros(5, 3))
        unused = torch.nn.Parameter(torch.zeros(7))                          # a parameter this step never touches on rank 1
        flat = D.FlatGradients(vols + [w, unused])
        assert flat.attached() and flat.padded % world == 0
        x = torch.full((1, 4, 6, 6, 6), float(rank + 1))
        loss = (vols[0] * x).sum() + (vols[1] * 2.0).sum() * (rank + 1) + (w * (rank + 1)).sum()
        if rank == 0:
            loss = loss + unused.sum()
        loss.backward()                                                       # accumulates INTO the flat buffer (the .grad views)
        assert flat.attached()
        flat.sync()
        shards = D.Shard()
        # lattice chunks: 5 chunks of 4 values, chunk c holds c everywhere; rank r owns c % world == r
        own = shards.chunks(5)
        local = torch.stack([torch.full((4,), float(c)) for c in own])
        lattice = shards.gather_chunks(local, 5)
        rows = shards.gather_rows(torch.full((D.ray_shard(7, rank, world)[1] - D.ray_shard(7, rank, world)[0], 2), float(rank)), 7)
        # the legacy helper with a missing gradient on one rank
        a, b = torch.nn.Parameter(torch.zeros(3)), torch.nn.Parameter(torch.zeros(2))
        a.grad = torch.full((3,), float(rank + 1))
        if rank == 0:
            b.grad = torch.full((2,), 4.0)
        never = torch.nn.Parameter(torch.zeros(4))                            # no rank has a gradient for it: it must keep None (Adam skips it)
        D.allreduce_gradients([a, never, b])
        assert never.grad is None
        if rank == 0:
            q.put(tuple(t.detach().numpy().copy() for t in (vols[0].grad, vols[1].grad, w.grad, unused.grad, lattice, rows, a.grad, b.grad)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_flat_gradient_exchange_lattice_chunks_and_missing_gradients_gloo():
    """Fine-tune exchange on the persistent flat buffer (reduce_scatter + all_gather on RCCL; all_reduce on gloo), K11's chunk-mod-world
    lattice sharding, the ray-shard gather, and the all-reduce helper when one rank has no gradient for a parameter (world size 2)."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_flat, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    v0, v1, w, unused, lattice, rows, a, b = (torch.from_numpy(x) for x in q.get(timeout=200))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert torch.allclose(v0, torch.full_like(v0, 1.5)) and torch.allclose(v1, torch.full_like(v1, 3.0)) and torch.allclose(w, torch.full_like(w, 1.5))
    assert torch.allclose(unused, torch.full_like(unused, 0.5))                     # mean of (1, nothing)
    assert torch.equal(lattice, torch.arange(5.0)[:, None].expand(5, 4))
    assert torch.equal(rows[:, 0], torch.tensor([0.0, 0, 0, 0, 1, 1, 1]))
    assert torch.allclose(a, torch.full_like(a, 1.5)) and torch.allclose(b, torch.full_like(b, 2.0))


def test_gradient_exchange_rejects_empty_and_mixed_parameter_lists():
    with pytest.raises(ValueError, match="no parameter"):
        D.FlatGradients([torch.nn.Parameter(torch.zeros(3), requires_grad=False)])
    with pytest.raises(ValueError, match="one dtype"):
        D.FlatGradients([torch.nn.Parameter(torch.zeros(3)), torch.nn.Parameter(torch.zeros(3, dtype=torch.float64))])
    groups = [{"params": [torch.nn.Parameter(torch.zeros(2))], "lr": 1.0}, {"params": torch.nn.Parameter(torch.zeros(3)), "lr": 2.0}]
    assert [t.numel() for t in D.optim_tensors(groups)] == [2, 3]


def test_single_process_shard_stand_in_collects_in_rank_order():
    sink = {}
    outs = [D.Shard.single(r, 3, sink).gather_rows(torch.full((D.ray_shard(8, r, 3)[1] - D.ray_shard(8, r, 3)[0], 1), float(r)), 8) for r in range(3)]
    assert outs[0] is None and outs[1] is None and torch.equal(outs[2][:, 0], torch.tensor([0.0, 0, 0, 1, 1, 1, 2, 2]))
    sink = {}
    got = None
    for r in range(3):
        sh = D.Shard.single(r, 3, sink)
        got = sh.gather_chunks(torch.stack([torch.full((2,), float(c)) for c in sh.chunks(7)]), 7)
    assert torch.equal(got, torch.arange(7.0)[:, None].expand(7, 2))


def _worker_eight(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        shards = D.Shard()
        n_rays = 24 * 32 + 5                                                  # not a multiple of the world size: shard lengths differ by one
        s, e = shards.rays(n_rays)
        # every rank draws the jitter of EVERY ray from the same generator state and renders its own slice (validate's contract); the "render"
        # here is a function of (ray index, jitter) only, so the gathered image must not depend on the partition
        torch.manual_seed(77)
        jitter = torch.rand(n_rays, 1)
        local = torch.cat([torch.arange(s, e, dtype=torch.float32)[:, None], jitter[s:e], (jitter[s:e] * 3.0).sin()], 1)
        image = shards.gather_rows(local, n_rays)
        own = shards.chunks(19)                                               # 19 lattice chunks over 8 ranks: three ranks own three, five own two
        lattice = shards.gather_chunks(torch.stack([torch.full((4,), float(c)) for c in own]), 19)
        flag = shards.any(rank == 5)                                          # one rank's overflow flag reaches everybody
        none = shards.any(False)
        p = torch.nn.Parameter(torch.zeros(1, 4, 5, 5, 5))                    # 500 floats: the flat buffer is padded to a multiple of 8
        flat = D.FlatGradients([p])
        (p * float(rank + 1)).sum().backward()
        flat.sync()
        q.put((rank, (s, e), image.numpy().copy(), lattice.numpy().copy(), flag, none, float(p.grad.mean()), flat.padded % world))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_eight_ranks_gather_the_same_image_lattice_and_gradients_gloo():
    """World size EIGHT (the node BASELINE names) on gloo: contiguous ray ranges that differ by at most one ray, the gathered (P, C) buffer
    identical on every rank and equal to the unsharded one, lattice chunks `index mod 8`, the one-flag all-reduce, the flat gradient exchange."""
    world, port = 8, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_eight, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=400) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    n_rays = 24 * 32 + 5
    torch.manual_seed(77)
    jitter = torch.rand(n_rays, 1)
    whole = torch.cat([torch.arange(n_rays, dtype=torch.float32)[:, None], jitter, (jitter * 3.0).sin()], 1)
    spans = [g[1] for g in got]
    assert spans[0][0] == 0 and spans[-1][1] == n_rays and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    assert max(e - s for s, e in spans) - min(e - s for s, e in spans) == 1
    for rank, _, image, lattice, flag, none, gmean, pad in got:
        assert torch.equal(torch.from_numpy(image), whole), rank              # bit for bit, on every rank
        assert torch.equal(torch.from_numpy(lattice), torch.arange(19.0)[:, None].expand(19, 4)), rank
        assert flag is True and none is False
        assert abs(gmean - 4.5) < 1e-6 and pad == 0                           # mean of 1 .. 8


def test_bench_ray_shards_of_the_headline_image_are_balanced_at_eight_ranks():
    """bench.py --gpus 8 --shard rays: every rank renders 38 400 of the 307 200 rays in two chunks of 19 200 (not 32 768 + 5 632)."""
    import importlib
    bench = importlib.import_module("bench")
    for rank in range(8):
        s, e = D.ray_shard(480 * 640, rank, 8)
        assert e - s == 38400
        chunk = bench.balanced_chunk(e - s, 32768)
        assert [min(chunk, e - s - o) for o in range(0, e - s, chunk)] == [19200, 19200]
