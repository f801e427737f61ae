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
