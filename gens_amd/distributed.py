"""Multi-GPU host logic: one process per GPU, torch.distributed ('nccl' == RCCL over xGMI on
ROCm; 'gloo' in the CPU tests).  The path shards embarrassingly -- rays of a scene, or whole scenes -- so the only
data-path collective is the gather of rendered buffers; training additionally all-reduces gradients (DDP in the
reference's runner.py:102-105, unchanged; `allreduce_gradients` below is the explicit form for fine-tune volumes).
"""
import torch
import torch.distributed as dist


def ray_shard(n_rays, rank, world):
    """Contiguous ray range [start, end) of `rank`; ranges differ in length by at most one ray."""
    base, rem = divmod(n_rays, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def scene_shard(n_scenes, rank, world):
    """Round-robin scene indices of `rank` (what DistributedSampler does in datasets/__init__.py:32-33)."""
    return list(range(rank, n_scenes, world))


def gather_rows(local, n_total, group=None):
    """all_gather of row-sharded (n_local, C) buffers whose shards follow `ray_shard` -> (n_total, C) on every rank."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    longest = -(-n_total // world)
    pad = torch.zeros(longest, *local.shape[1:], dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    out = []
    for r, p in enumerate(parts):
        s, e = ray_shard(n_total, r, world)
        out.append(p[:e - s])
    assert out[rank].shape[0] == local.shape[0]
    return torch.cat(out, 0)


def render_sharded(render_fn, rays_o, rays_d, jitter=None, group=None):
    """Render a scene's rays split across the ranks of `group` and gather the per-ray buffers on every rank.

    render_fn(rays_o, rays_d, jitter_slice) -> dict of per-ray tensors (n_local, ...) on the local device.
    `jitter` (n_rays, 1) must be the SAME tensor on every rank (draw it from an identically seeded generator): it is
    sliced with the rays, which makes the image independent of the partition.
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n = rays_o.shape[0]
    s, e = ray_shard(n, rank, world)
    local = render_fn(rays_o[s:e], rays_d[s:e], None if jitter is None else jitter[s:e])
    return {k: gather_rows(v.reshape(v.shape[0], -1), n, group).reshape(n, *v.shape[1:]) for k, v in local.items()}


def allreduce_gradients(params, group=None, average=True):
    """Bucket-free gradient all-reduce for a short parameter list (fine-tune: MLPs + the volume pyramid).
    Gradients are flattened into ONE buffer so a single large collective crosses xGMI (few, large messages)."""
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, group=group)
    if average:
        flat /= dist.get_world_size(group)
    off = 0
    for g in grads:
        g.copy_(flat[off:off + g.numel()].reshape(g.shape))
        off += g.numel()
