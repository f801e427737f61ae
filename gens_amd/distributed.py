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


def _all_gather_equal(pad, group):
    """all_gather of equally shaped buffers -> list of per-rank tensors.  On RCCL ('nccl') one all_gather_into_tensor, device to device;
    gloo (CPU tests) has no flat form for every dtype, so it takes the list form."""
    world = dist.get_world_size(group)
    if dist.get_backend(group) == "nccl":
        flat = torch.empty(world * pad.shape[0], *pad.shape[1:], dtype=pad.dtype, device=pad.device)
        dist.all_gather_into_tensor(flat, pad, group=group)
        return list(flat.reshape(world, *pad.shape).unbind(0))
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return parts


def gather_rows(local, n_total, group=None):
    """all_gather of row-sharded (n_local, C) buffers whose shards follow `ray_shard` -> (n_total, C) on every rank."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    longest = -(-n_total // world)
    pad = torch.zeros(longest, *local.shape[1:], dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    parts = _all_gather_equal(pad, group)
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


class Shard:
    """This process' place in a ray- / lattice-sharded evaluation of ONE scene (BASELINE config 4: "ray batches sharded across 8 x MI355X,
    RCCL gather of rendered buffers"; the loops it splits: implicit_surface.py:407-427, 437-453).
    `Shard.single(rank, world, sink)` is the collective-free stand-in the one-GPU tests use: shards are rendered one after the other in
    one process and `sink` collects them."""

    def __init__(self, group=None):
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self._sink = None

    @classmethod
    def single(cls, rank, world, sink):
        self = cls.__new__(cls)
        self.group, self.rank, self.world, self._sink = None, rank, world, sink
        return self

    def rays(self, n_rays):
        return ray_shard(n_rays, self.rank, self.world)

    def gather_rows(self, local, n_total, key="rows"):
        """(n_local, C) row shard -> the (n_total, C) buffer of all ranks."""
        if self._sink is not None:
            self._sink.setdefault(key, {})[self.rank] = local.clone()
            if len(self._sink[key]) < self.world:
                return None
            return torch.cat([self._sink[key][r] for r in range(self.world)], 0)
        return gather_rows(local, n_total, self.group)

    def any(self, flag):
        """True on every rank if `flag` is true on any rank (one tiny all-reduce); the single-process stand-in has nobody to ask."""
        if self._sink is not None or self.world == 1:
            return bool(flag)
        dev = "cuda" if dist.get_backend(self.group) == "nccl" else "cpu"
        t = torch.tensor([1.0 if flag else 0.0], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return bool(t.item() > 0)

    def chunks(self, n_chunks):
        """Lattice chunks of this rank: chunk index mod world."""
        return list(range(self.rank, n_chunks, self.world))

    def gather_chunks(self, local, n_chunks, key="chunks"):
        """local (n_own, chunk_len): this rank's chunks in increasing index order -> (n_chunks, chunk_len) on every rank."""
        per = -(-n_chunks // self.world)
        pad = torch.zeros(per, local.shape[1], dtype=local.dtype, device=local.device)
        pad[:local.shape[0]] = local
        if self._sink is not None:
            self._sink.setdefault(key, {})[self.rank] = pad
            if len(self._sink[key]) < self.world:
                return None
            parts = [self._sink[key][r] for r in range(self.world)]
        else:
            parts = _all_gather_equal(pad, self.group)
        # chunk c lives at parts[c % world][c // world]
        return torch.stack(parts, 1).reshape(per * self.world, -1)[:n_chunks]


def allreduce_gradients(params, group=None, average=True):
    """Bucket-free gradient all-reduce for a short parameter list (fine-tune: MLPs + the volume pyramid).
    Gradients are flattened into ONE buffer so a single large collective crosses xGMI (few, large messages).  Every parameter that
    requires grad takes part -- a missing gradient (an unused head, a rank without pseudo points) counts as zeros -- so the flat layout
    is the same on every rank whatever each rank's step touched."""
    params = [p for p in params if p.requires_grad]
    if not params:
        return
    _check_uniform(params)
    # one presence flag per parameter rides at the end of the same buffer: a parameter whose gradient is None on EVERY rank keeps None
    # (Adam skips it, as it does on one GPU and under the reference's DDP); one that any rank produced becomes dense everywhere
    present = torch.tensor([0.0 if p.grad is None else 1.0 for p in params], dtype=params[0].dtype, device=params[0].device)
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params] + [present])
    dist.all_reduce(flat, group=group)
    n = flat.numel() - len(params)
    seen = (flat[n:] > 0).tolist()
    if average:
        flat[:n] /= dist.get_world_size(group)
    off = 0
    for p, any_rank in zip(params, seen):
        g = flat[off:off + p.numel()].reshape(p.shape)
        off += p.numel()
        if not any_rank:
            continue
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)


def _check_uniform(params):
    p0 = params[0]
    for p in params:
        if p.dtype != p0.dtype or p.device != p0.device:
            raise ValueError(f"gradient exchange needs parameters of one dtype on one device: got {p.dtype} on {p.device} next to {p0.dtype} on {p0.device}")


class FlatGradients:
    """Fine-tune data parallelism (BASELINE config 5; the reference would wrap the model in DDP, runner.py:102-105): the gradients of the
    volume pyramid (307 MB at five levels) and the MLPs live in ONE persistent flat buffer -- `p.grad` of every parameter is a view of
    it, so autograd accumulates straight into the communication buffer (no cat, no copy back) -- and a step's exchange is
    reduce_scatter + all_gather on that buffer: each rank reduces 1/world of it and every byte crosses each xGMI link once, where the
    ring all-reduce of a dense `torch.cat` copy made three more passes over the 307 MB and paid the concatenation.

        flat = FlatGradients(model.get_optim_params(...)' tensors)      # once
        loss.backward(); flat.sync(); optimizer.step(); flat.zero()      # per step (do NOT zero_grad(set_to_none=True))"""

    def __init__(self, params, group=None):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("FlatGradients: no parameter requires a gradient")
        _check_uniform(self.params)
        self.group = group
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        n = sum(p.numel() for p in self.params)
        self.numel = n
        self.padded = -(-n // world) * world
        p0 = self.params[0]
        self.flat = torch.zeros(self.padded, dtype=p0.dtype, device=p0.device)
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view(p.shape)
            off += p.numel()

    def zero(self):
        self.flat.zero_()

    def attached(self):
        """True while every parameter's .grad still is its view of the flat buffer (zero_grad(set_to_none=True) detaches them)."""
        off = 0
        for p in self.params:
            if p.grad is None or p.grad.data_ptr() != self.flat.data_ptr() + off * self.flat.element_size():
                return False
            off += p.numel()
        return True

    def sync(self, average=True):
        assert self.attached(), "a parameter's .grad no longer aliases the flat buffer (use FlatGradients.zero(), not zero_grad(set_to_none=True))"
        if not dist.is_initialized() or dist.get_world_size(self.group) == 1:
            return
        world = dist.get_world_size(self.group)
        shard = torch.empty(self.padded // world, dtype=self.flat.dtype, device=self.flat.device)
        if dist.get_backend(self.group) == "nccl":
            dist.reduce_scatter_tensor(shard, self.flat, group=self.group)
            if average:
                shard /= world
            dist.all_gather_into_tensor(self.flat, shard, group=self.group)
        else:                                   # gloo (CPU tests): no reduce_scatter; same result through all_reduce
            dist.all_reduce(self.flat, group=self.group)
            if average:
                self.flat /= world


def optim_tensors(param_groups):
    """The tensors of Adam-style parameter groups ({"params": tensor or list, ...}, GenS.get_optim_params) in group order."""
    out = []
    for g in param_groups:
        ps = g["params"]
        out += [ps] if isinstance(ps, torch.Tensor) else list(ps)
    return out


class FinetuneStepper:
    """One data-parallel fine-tune step after another (BASELINE config 5; the loop it replaces: runner.py:294-316 with the model wrapped in
    DDP, :102-105): every rank renders ITS rays of the same scene, the gradients of the volume pyramid and the MLPs meet in the flat
    buffer (FlatGradients: reduce_scatter + all_gather over RCCL), every rank takes the same optimiser step.

        stepper = FinetuneStepper(model, optimizer, loss_fn)          # after model.init_volumes(...) and get_optim_params(...)
        loss = stepper.step(ipts, cos_anneal_ratio, step)             # ipts: this rank's rays of the step

    Differences from DDP, both documented behaviour: gradients are dense (a parameter a step does not touch gets zeros, not None -- with
    Adam its moments decay instead of standing still; under DDP without find_unused_parameters such a step is an error), and they must be
    cleared with FlatGradients.zero(), which step() does."""

    def __init__(self, model, optimizer, loss_fn, group=None):
        self.model, self.optimizer, self.loss_fn = model, optimizer, loss_fn
        self.flat = FlatGradients(optim_tensors(optimizer.param_groups), group)

    def step(self, ipts, cos_anneal_ratio=1.0, step=None):
        self.flat.zero()
        outputs = self.model("finetune", ipts, cos_anneal_ratio=cos_anneal_ratio, step=step)
        loss = self.loss_fn(outputs, ipts)
        loss.backward()
        self.flat.sync()
        # what the reference raises inside forward() (no valid pseudo point, a singular camera: implicit_surface.py:494-495) the fused step leaves
        # as device flags: look at them BEFORE the update is applied (waits for this step's forward only; the backward is already enqueued)
        for m in self.model.modules():
            if hasattr(m, "check_deferred"):
                m.check_deferred()
        self.optimizer.step()
        return loss.detach(), outputs
