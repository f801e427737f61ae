"""Multi-GPU training wiring on the one GPU a test box has (runner.py:102-105):

  * two SPAWNED ranks sharing device 0 on gloo, the model wrapped in DistributedDataParallel exactly as runner.py:104 wraps it (no
    find_unused_parameters): after one train step every rank holds the MEAN of the two ranks' gradients;
  * world size 1 on 'nccl' (= RCCL): the wrapped model's step equals the unwrapped one's bit for bit;
  * FinetuneStepper (FlatGradients: reduce_scatter + all_gather on one persistent flat buffer) drives fine-tune steps the same way.
The first test starts worker processes, which a process that has initialised the GPU must not do (HIP does not survive a fork, and the GPU
boxes refuse an exec from such a process): tests/conftest.py orders it before every test that touches the GPU, and its own check only counts
devices, which initialises nothing."""
import os
import socket

import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class TinyFeatureNet(nn.Module):
    def __init__(self, confs):
        super().__init__()
        self.convs = nn.ModuleList([nn.Conv2d(3, 4, 3, padding=1) for _ in range(5)])

    def forward(self, imgs):
        outs, x = [], imgs
        for conv in self.convs:
            outs.append(conv(x))
            x = nn.functional.avg_pool2d(x, 2)
        return outs


class TinyRegNet(nn.Module):
    def __init__(self, confs):
        super().__init__()
        self.convs = nn.ModuleList([nn.Conv3d(8, 4, 1) for _ in confs.get_list("d_out")])

    def forward(self, volumes):
        return [conv(v) for conv, v in zip(self.convs, volumes)]


def _model():
    from gens_amd.config import gens_model_conf
    from gens_amd.models import gens
    gens.register_backbones(TinyFeatureNet, TinyRegNet)
    torch.manual_seed(0)
    return gens.GenS(gens_model_conf(volume_dims=(16, 8, 4))).cuda().train()


def _inputs(seed, n_rays=48, nv=4):
    from gens_amd import synthetic
    sc = synthetic.make_scene(nv=nv, h=48, w=64, n_levels=1, seed=5)
    g = torch.Generator().manual_seed(seed)
    pix = torch.stack([torch.randint(4, 60, (n_rays,), generator=g), torch.randint(4, 44, (n_rays,), generator=g)], -1)
    ro, rd = synthetic.make_rays(sc["intrs"], sc["c2ws"], 48, 64, pixels=pix)
    ipts = {"imgs": sc["imgs"], "intrs": sc["intrs"], "c2ws": sc["c2ws"], "rays_o": ro, "rays_d": rd, "near": sc["near"], "far": sc["far"],
            "pseudo_pts": torch.rand(256, 3, generator=g) - 0.5}
    return {k: v.cuda() for k, v in ipts.items()}


def _loss(out, _ipts=None):
    hit = out["mid_inside_sphere"].reshape(1, -1, 1, 1)
    mfc = (((out["sampled_gray_val"] - out["ref_gray_val"]) ** 2) * hit).mean()
    loss = (out["color_fine"].abs().sum() + 0.1 * out["gradient_error"] + 0.01 * out["smooth_error"] + 0.01 * out["tv_reg"]
            + torch.exp(-out["sparse_sdf"].abs() * 100).mean() + mfc + 0.1 * out["render_depth"].sum())
    return loss + out["pseudo_sdf"].abs().mean() if "pseudo_sdf" in out else loss


def _train_grads(model, seed):
    """One train step's gradients of `model` (possibly DDP-wrapped) on the batch of `seed`: {name: tensor on the host}."""
    for p in model.parameters():
        p.grad = None
    torch.manual_seed(100 + seed)                      # the two host-generator draws of a step (implicit_surface.py:256,362)
    out = model("train", _inputs(seed), cos_anneal_ratio=0.5, step=1.0)
    _loss(out).backward()
    bare = model.module if hasattr(model, "module") else model
    return {k: p.grad.detach().cpu().clone() for k, p in bare.named_parameters() if p.grad is not None}


def _ddp_rank(rank, world, port, q):
    """A spawned rank: both share GPU 0, gloo carries the gradient all-reduce (RCCL wants one device per rank)."""
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        model = _model()
        ddp = DistributedDataParallel(model, device_ids=[0])            # runner.py:104, literally
        got = _train_grads(ddp, seed=rank)
        want = None
        if rank == 0:                                                    # the mean of the two batches' gradients, without DDP
            a, b = _train_grads(model, seed=0), _train_grads(model, seed=1)
            want = {k: 0.5 * (a[k] + b[k]) for k in a}
        dist.barrier()
        dist.destroy_process_group()
        as_np = lambda d: None if d is None else {k: v.numpy() for k, v in d.items()}  # noqa: E731   (plain pickles: no shared-memory hand-over)
        q.put((rank, None, as_np(got), as_np(want)))
    except Exception as e:                                               # a dead worker must not hang the parent
        import traceback
        q.put((rank, f"{type(e).__name__}: {e}\n{traceback.format_exc()}", None, None))


@pytest.mark.forks_before_gpu
def test_ddp_wrapped_gens_two_ranks_on_one_gpu_average_their_gradients():
    if torch.cuda.is_initialized():
        pytest.skip("the GPU is already initialised in this process: starting worker processes is no longer allowed (run this test first or alone)")
    import multiprocessing as mp
    ctx = mp.get_context("spawn")                                        # fresh interpreters: each initialises the GPU itself
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_rank, args=(r, 2, port, q)) for r in range(2)]
    try:
        for p in procs:
            p.start()
    except OSError as e:                                                 # a box that refuses to start programs from this process
        for p in procs:
            if p.is_alive():
                p.kill()
        pytest.skip(f"cannot start worker processes here: {e}")
    res = {}
    try:
        for _ in procs:
            rank, err, got, want = q.get(timeout=600)
            assert err is None, err
            as_t = lambda d: None if d is None else {k: torch.from_numpy(v) for k, v in d.items()}  # noqa: E731
            res[rank] = (as_t(got), as_t(want))
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    want = res[0][1]
    assert set(res[0][0]) == set(res[1][0]) == set(want) and len(want) > 40
    top = max(float(v.abs().max()) for v in want.values())
    for k, w in want.items():
        for r in (0, 1):
            err = float((res[r][0][k] - w).abs().max())
            assert err <= 2e-5 * max(float(w.abs().max()), 1e-3 * top), (k, r, err)
    for k in want:                                                       # every rank ends the step with the SAME gradients
        assert torch.equal(res[0][0][k], res[1][0][k]), k


def test_ddp_wrapped_gens_world_size_one_on_rccl_matches_the_unwrapped_model():
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1")
    dist.init_process_group("nccl", rank=0, world_size=1)               # 'nccl' is RCCL on ROCm
    try:
        model = _model()
        want = _train_grads(model, seed=3)
        ddp = DistributedDataParallel(model, device_ids=[0])
        got = _train_grads(ddp, seed=3)
        assert set(got) == set(want)
        top = max(float(v.abs().max()) for v in want.values())
        for k in want:                                                   # (float atomics in the scatter kernels: equal to round-off)
            assert float((got[k] - want[k]).abs().max()) <= 2e-5 * max(float(want[k].abs().max()), 1e-3 * top), k
        # every parameter that requires a gradient received one: DDP without find_unused_parameters (runner.py:104) needs exactly that
        missing = [k for k, p in model.named_parameters() if p.requires_grad and k not in got]
        assert not missing, missing
        ddp("train", _inputs(4), cos_anneal_ratio=0.5, step=2.0)        # a second step through the same reducer
    finally:
        dist.destroy_process_group()


def test_finetune_stepper_drives_flat_gradients():
    """FinetuneStepper (gens_amd/distributed.py): fine-tune steps whose gradients live in ONE flat buffer; world size 1 on RCCL, so the
    exchange is the identity and the step must equal a plain zero_grad / backward / Adam step bit for bit in its first iteration."""
    import torch.distributed as dist
    from gens_amd.distributed import FinetuneStepper, optim_tensors
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1")
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        model = _model()
        ipts = _inputs(7, nv=3)
        model.init_volumes({k: ipts[k] for k in ("imgs", "intrs", "c2ws")})
        ipts["view_ids"] = [0, 1, 2]
        ref = _model()                                                   # the same seed: the same weights (weight-normed modules do not deepcopy)
        ref.load_state_dict(model.state_dict(), strict=False)
        ref.init_volumes({k: ipts[k] for k in ("imgs", "intrs", "c2ws")})
        for a, b in zip(model.volumes, ref.volumes):
            assert torch.equal(a, b)
        lrs = {"mlp_lr": 5e-4, "vol_lr": [1e-2, 1e-2, 1e-2]}
        opt = torch.optim.Adam(model.get_optim_params(lrs))
        stepper = FinetuneStepper(model, opt, _loss)
        assert stepper.flat.attached() and stepper.flat.numel == sum(p.numel() for p in optim_tensors(opt.param_groups) if p.requires_grad)
        torch.manual_seed(11)
        loss, out = stepper.step(ipts, 1.0, None)
        ref_opt = torch.optim.Adam(ref.get_optim_params(lrs))
        torch.manual_seed(11)
        ref_loss = _loss(ref("finetune", ipts, cos_anneal_ratio=1.0, step=None))
        ref_opt.zero_grad()
        ref_loss.backward()
        ref_opt.step()
        assert abs(float(loss) - float(ref_loss)) <= 1e-5 * abs(float(ref_loss))
        for (k, a), (_, b) in zip(model.named_parameters(), ref.named_parameters()):
            if a.requires_grad:
                assert float((a - b).abs().max()) <= 1e-5 * max(float(b.abs().max()), 1e-3), k
        torch.manual_seed(12)
        stepper.step(ipts, 1.0, None)                                    # the views survive an optimiser step and a zero()
        assert stepper.flat.attached()
    finally:
        dist.destroy_process_group()
