"""Multi-GPU training wiring on the one GPU a test box has (runner.py:102-105):

  * two SPAWNED ranks sharing device 0 on gloo, the model wrapped in DistributedDataParallel exactly as runner.py:104 wraps it (no
    find_unused_parameters): after one train step every rank holds the MEAN of the two ranks' gradients;
  * world size 1 on 'nccl' (= RCCL): the wrapped model's step equals the unwrapped one's bit for bit;
  * FinetuneStepper (FlatGradients: reduce_scatter + all_gather on one persistent flat buffer) drives fine-tune steps the same way;
  * round 6 -- the CAPTURED step under data parallelism: runner.py:157-166's loop run for seven steps with the model wrapped in
    DistributedDataParallel (two ranks on gloo, one on RCCL) and through FinetuneStepper, AutoGraph on: after two eager steps forward and
    backward are HIP-graph replays, the reducer's hooks fire behind an autograd.Function whose backward IS a replay, FlatGradients' .grad views
    take the replayed gradients -- and the parameters walk the trajectory of the un-wrapped, un-captured loop on the averaged batches, the same
    on every rank after every step.  A capture that fails under DDP leaves the loop on the eager path of the same process.
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


# ----------------------------------------------------------------------------------------------------------------------
# Round 6: the captured step (gens_amd.graph.AutoGraph) under data parallelism, for more than the one or two steps the tests above run
# ----------------------------------------------------------------------------------------------------------------------
# rgb_fc.4.bias shifts every view's score alike: the soft-max over views does not see it, its gradient is analytically ZERO -- 1e-8 of round-off in
# any implementation, which Adam's normalisation turns into steps of +-lr whose signs follow the summation order (here: the order the ranks'
# gradients are added in).  Every other parameter is compared.
NOISE_ONLY = "color_network.rgb_fc.4.bias"
N_LOOP = 7          # two eager warm-up steps, one capture, five replays (the capture step itself replays)
TRAIN_LRS = {"mlp_lr": 5e-4, "feat_lr": 1e-3}
FT_LRS = {"mlp_lr": 5e-4, "vol_lr": [1e-2, 1e-2, 1e-2]}


def _loop_batch(k, rank, finetune):
    """The batch of (step k, rank): its own rays and pseudo points (the DistributedSampler's share / this rank's rays of the fine-tune scene)."""
    ipts = _inputs(1000 + 16 * k + rank, nv=3 if finetune else 4)
    if finetune:
        ipts["view_ids"] = [[0, 1, 2], [1, 0, 2], [2, 1, 0]][k % 3]
    return ipts


def _loop_call(k):
    return {"cos_anneal_ratio": min(1.0, 0.25 * k), "step": 1.0 + k / 16}


def _digest(model):
    import hashlib
    h = hashlib.sha1()
    for _, p in sorted(model.named_parameters()):
        h.update(p.detach().cpu().numpy().tobytes())
    return h.hexdigest()


def _params_np(model):
    return {k: p.detach().cpu().numpy().copy() for k, p in model.named_parameters() if p.requires_grad}


def _finetune_ready(auto):
    model = _model()
    ipts = _inputs(7, nv=3)
    model.init_volumes({k: ipts[k] for k in ("imgs", "intrs", "c2ws")})
    model.auto_graph = auto
    return model


def _ddp_train_loop(rank, world, wrap):
    """runner.py:96-104, 157-166 as written: Adam over get_optim_params, the model wrapped, then model(...), loss, zero_grad, backward, step, the
    loss read back.  -> losses, parameter digest after every step, final parameters, AutoGraph statistics."""
    model = _model()
    opt = torch.optim.Adam(model.get_optim_params(TRAIN_LRS))
    call = wrap(model)
    losses, digests = [], []
    for k in range(N_LOOP):
        ipts = _loop_batch(k, rank, False)
        torch.manual_seed(7000 + 16 * k + rank)            # the step's host-generator draws (implicit_surface.py:256,362)
        out = call("train", ipts, **_loop_call(k))
        loss = _loss(out)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss))
        digests.append(_digest(model))
    return losses, digests, _params_np(model), dict(model._auto.stats)


def _stepper_loop(rank, world):
    """FinetuneStepper over FlatGradients, AutoGraph on.  -> losses, digests, final parameters, statistics, flat.attached() after the last step."""
    from gens_amd.distributed import FinetuneStepper
    model = _finetune_ready(True)
    opt = torch.optim.Adam(model.get_optim_params(FT_LRS))
    stepper = FinetuneStepper(model, opt, _loss)
    losses, digests = [], []
    for k in range(N_LOOP):
        ipts = _loop_batch(k, rank, True)
        torch.manual_seed(9000 + 16 * k + rank)
        loss, _ = stepper.step(ipts, **_loop_call(k))
        losses.append(float(loss))
        digests.append(_digest(model))
        assert stepper.flat.attached(), k                  # the .grad views survive a replay, the exchange, the update and zero()
    return losses, digests, _params_np(model), dict(model._auto.stats), stepper.flat.attached()


def _reference_loop(world, finetune):
    """The trajectory the data-parallel runs must walk: ONE process, the bare model, every call eager (auto_graph off), the ranks' batches one after
    the other, their gradients averaged the way DDP / FlatGradients average them (each divided by the world size, then summed in rank order)."""
    model = _finetune_ready(False) if finetune else _model()
    model.auto_graph = False
    opt = torch.optim.Adam(model.get_optim_params(FT_LRS if finetune else TRAIN_LRS))
    seed0 = 9000 if finetune else 7000
    losses = [[] for _ in range(world)]
    named = [(k, p) for k, p in model.named_parameters() if p.requires_grad]
    for k in range(N_LOOP):
        mean = {}
        for r in range(world):
            for _, p in named:
                p.grad = None
            torch.manual_seed(seed0 + 16 * k + r)
            out = model("finetune" if finetune else "train", _loop_batch(k, r, finetune), **_loop_call(k))
            loss = _loss(out)
            loss.backward()
            losses[r].append(float(loss))
            for name, p in named:
                g = (p.grad if p.grad is not None else torch.zeros_like(p)) / world
                mean[name] = g if r == 0 else mean[name] + g
        for name, p in named:
            p.grad = mean[name]
        opt.step()
    model.implicit_surface.check_deferred()
    return losses, _params_np(model)


def _check_against_reference(got_by_rank, ref_losses, ref_params, rel, prel):
    """got_by_rank: {rank: (losses, digests, params, stats, ...)}"""
    import numpy as np
    world = len(got_by_rank)
    for r in range(world):
        losses, digests, params, stats = got_by_rank[r][:4]
        assert stats["captured"] >= 1 and stats["replayed"] >= 3 and stats["eager"] == 2, (r, stats)
        assert len(set(losses)) == N_LOOP, losses             # different steps, not one step replayed
        for a, b in zip(losses, ref_losses[r]):
            assert abs(a - b) <= rel * abs(b), (r, losses, ref_losses[r])
        assert set(params) == set(ref_params)
        for k, v in params.items():
            if k.endswith(NOISE_ONLY):
                continue
            ref = ref_params[k]
            assert float(np.abs(v - ref).max()) <= prel * max(float(np.abs(ref).max()), 1e-2), (r, k)
        assert digests == got_by_rank[0][1], r                # every rank ends EVERY step with the same parameters, bit for bit


def _captured_rank(rank, world, port, q):
    """A spawned rank of the two-rank runs: both share GPU 0, gloo carries the exchange."""
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        train = _ddp_train_loop(rank, world, lambda m: DistributedDataParallel(m, device_ids=[0]))      # runner.py:104, literally
        dist.barrier()
        ft = _stepper_loop(rank, world)
        dist.barrier()
        dist.destroy_process_group()
        ref = None
        if rank == 0:
            ref = (_reference_loop(world, False), _reference_loop(world, True))
        q.put((rank, None, train, ft, ref))
    except Exception as e:                                               # a dead worker must not hang the parent
        import traceback
        q.put((rank, f"{type(e).__name__}: {e}\n{traceback.format_exc()}", None, None, None))


@pytest.mark.forks_before_gpu
def test_captured_steps_under_ddp_and_flat_gradients_two_ranks_walk_the_reference_trajectory():
    if torch.cuda.is_initialized():
        pytest.skip("the GPU is already initialised in this process: starting worker processes is no longer allowed (run this test first or alone)")
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_captured_rank, args=(r, 2, port, q)) for r in range(2)]
    try:
        for p in procs:
            p.start()
    except OSError as e:
        for p in procs:
            if p.is_alive():
                p.kill()
        pytest.skip(f"cannot start worker processes here: {e}")
    res = {}
    try:
        for _ in procs:
            rank, err, train, ft, ref = q.get(timeout=900)
            assert err is None, err
            res[rank] = (train, ft, ref)
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    (ref_train_losses, ref_train_params), (ref_ft_losses, ref_ft_params) = res[0][2]
    _check_against_reference({r: res[r][0] for r in res}, ref_train_losses, ref_train_params, rel=1e-4, prel=2e-3)
    _check_against_reference({r: res[r][1] for r in res}, ref_ft_losses, ref_ft_params, rel=1e-4, prel=2e-3)
    assert all(res[r][1][4] for r in res)                                # FlatGradients still attached after the last step, on every rank


def test_captured_steps_under_ddp_world_size_one_on_rccl():
    """The same loops with RCCL as the backend (one rank: the exchange is the identity, the reducer, its hooks and its bucket kernels are all there)."""
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1")
    dist.init_process_group("nccl", rank=0, world_size=1)               # 'nccl' is RCCL on ROCm
    try:
        train = _ddp_train_loop(0, 1, lambda m: DistributedDataParallel(m, device_ids=[0]))
        ft = _stepper_loop(0, 1)
    finally:
        dist.destroy_process_group()
    ref_losses, ref_params = _reference_loop(1, False)
    _check_against_reference({0: train}, ref_losses, ref_params, rel=1e-4, prel=2e-3)
    ref_losses, ref_params = _reference_loop(1, True)
    _check_against_reference({0: ft}, ref_losses, ref_params, rel=1e-4, prel=2e-3)
    assert ft[4]


def test_a_failed_capture_under_ddp_stays_eager_in_the_same_process():
    """Something the capture cannot hold (a stream synchronisation inside the step) under a DDP wrapper: a warning, the signature stays eager, the
    reducer keeps averaging -- the loop walks the un-captured trajectory in THIS process (nothing is re-executed)."""
    import warnings
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1")
    dist.init_process_group("nccl", rank=0, world_size=1)
    pid = os.getpid()
    try:
        calls = []

        def wrap(model):
            inner = model._forward_impl

            def spoiled(*a, **k):
                calls.append(torch.cuda.is_current_stream_capturing())
                if calls[-1]:
                    torch.cuda.current_stream().synchronize()           # illegal while capturing
                return inner(*a, **k)
            model._forward_impl = spoiled
            return DistributedDataParallel(model, device_ids=[0])
        with warnings.catch_warnings(record=True) as seen:
            warnings.simplefilter("always")
            model_stats = None
            model = _model()
            opt = torch.optim.Adam(model.get_optim_params(TRAIN_LRS))
            call = wrap(model)
            losses = []
            for k in range(N_LOOP):
                torch.manual_seed(7000 + 16 * k)
                out = call("train", _loop_batch(k, 0, False), **_loop_call(k))
                loss = _loss(out)
                opt.zero_grad()
                loss.backward()
                opt.step()
                losses.append(float(loss))
            model_stats = dict(model._auto.stats)
        assert any("cannot be captured" in str(w.message) for w in seen), [str(w.message) for w in seen]
        assert calls.count(True) == 1 and os.getpid() == pid
        assert model_stats["captured"] == 0 and model_stats["replayed"] == 0 and model_stats["eager"] == N_LOOP, model_stats
    finally:
        dist.destroy_process_group()
    import numpy as np
    ref_losses, ref_params = _reference_loop(1, False)
    for a, b in zip(losses, ref_losses[0]):
        assert abs(a - b) <= 1e-4 * abs(b), (losses, ref_losses[0])
    for k, v in _params_np(model).items():
        if not k.endswith(NOISE_ONLY):
            assert float(np.abs(v - ref_params[k]).max()) <= 2e-3 * max(float(np.abs(ref_params[k]).max()), 1e-2), k
