#!/usr/bin/env python3
"""Timing of one training step of the hot path (BASELINE config 3 / 5 shape): 5 views 480x640, volume_dims 256/128/64, 512 rays + 2048 pseudo
points, the reference's Loss, backward into the SDF / colour MLPs, the volumes and the feature pyramid, Adam.

The loop is runner.py's (157-166 / 300-308): `outputs = model(...)`, the loss, `optimizer.zero_grad()`, `loss.backward()`, `optimizer.step()`, the
loss read back -- nothing in it knows about graphs.  What runs behind `model(...)`:
    (default)    the captured step of gens_amd.graph.AutoGraph: after two eager calls, forward and backward are one HIP graph replay each
    --no-auto    every call eager (GENS_AUTO_GRAPH=0: rounds 1 - 4's "eager" figures)
    --graph      the WHOLE step -- forward, loss, backward, optimiser -- captured by the caller (gens_amd.graph.GraphedStep): the lower bound
Workloads:
    (default)    "hot path": K1 + ImplicitSurface.forward("train") on leaf feature maps / regularised volumes (the CNNs' outputs stand-ins)
    --finetune   GenS(has_vol).forward("train") as runner.py:300 calls it: the volumes are the parameters (config 5); --conf-shape: 1152 x 1600,
                 three views, five levels (confs/gens_finetune.conf as shipped)
    --full       GenS.forward("train") with the 2-D CNN (twice), K1 and the 3-D U-Net inside (config 3 as runner.py runs it)
The optimiser of the GenS workloads is torch.optim.Adam(model.get_optim_params(lrs)) exactly as runner.py:96-97 builds it (get_optim_params asks for
the fused update in its groups; GENS_FUSED_ADAM=0 / --foreach-adam: the multi-tensor default)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gens_amd import ops, synthetic  # noqa: E402
from gens_amd.config import gens_loss_conf, gens_model_conf  # noqa: E402
from gens_amd.losses import Loss  # noqa: E402
from gens_amd.models.modules.implicit_surface import ImplicitSurface  # noqa: E402


def measure(argv=(), quiet=False, kernels=False):
    """One measurement; argv: the command-line flags below.  -> (milliseconds per step, label, table); table: with kernels=True
    {C-ABI entry: {"launches", "ms", "bytes", "flops"}} of ONE extra, untimed step, else {}."""
    saved = sys.argv
    sys.argv = [saved[0], *argv]
    try:
        got = _measure(quiet, kernels)
        return got if kernels else (*got, {})
    finally:
        sys.argv = saved


def main():
    _measure(False)


def _measure(quiet, kernels=False):
    dev = torch.device("cuda:0")
    if "--miopen-find" in sys.argv:                                     # the reference's own setting (runner.py:26): MIOpen searches per conv shape
        torch.backends.cudnn.benchmark = True
    if os.environ.get("GENS_BLAS"):                                     # "hipblaslt" / "cublas" (rocBLAS): which GEMM library torch uses
        torch.backends.cuda.preferred_blas_library(os.environ["GENS_BLAS"])
    conf_shape = "--conf-shape" in sys.argv     # confs/gens_finetune.conf:5-16,52-54 as shipped: img_hw [1152, 1600], num_views 3, five levels
    dims = [256, 128, 64, 32, 16] if ("--levels5" in sys.argv or conf_shape) else [256, 128, 64]       # --levels5: the shipped confs/gens.conf pyramid
    nv, h, w = (3, 1152, 1600) if conf_shape else (5, 480, 640)
    sc = synthetic.make_scene(nv=nv, h=h, w=w, n_levels=5, seed=0)
    imgs, intrs, c2ws = sc["imgs"].to(dev), sc["intrs"].to(dev), sc["c2ws"].to(dev)
    feats = [f.to(dev).requires_grad_(True) for f in sc["features"]]
    vols = [v.to(dev).requires_grad_(True) for v in synthetic.make_volumes(dims, seed=1)]
    torch.manual_seed(0)
    surf = ImplicitSurface(gens_model_conf(volume_dims=tuple(dims))["implicit_surface"]).to(dev).train()
    g = torch.Generator().manual_seed(3)
    pix = torch.stack([torch.randint(0, w, (512,), generator=g), torch.randint(0, h, (512,), generator=g)], -1)
    ro, rd = synthetic.make_rays(sc["intrs"], sc["c2ws"], h, w, pixels=pix)
    ipts = {"imgs": imgs, "intrs": intrs, "c2ws": c2ws, "rays_o": ro.to(dev), "rays_d": rd.to(dev), "near": sc["near"].to(dev),
            "far": sc["far"].to(dev), "pseudo_pts": (torch.rand(2048, 3, generator=g) - 0.5).to(dev)}
    target = torch.rand(512, 3, device=dev)
    targets = {"color": target}
    # the reference's Loss module with the shipped weights (confs/gens.conf:47-59, confs/gens_finetune.conf:32-41), as runner.py:99,161 uses it
    train_loss, ft_loss = Loss(gens_loss_conf()).to(dev), Loss(gens_loss_conf(finetune=True)).to(dev)
    if "--freeze-color" in sys.argv:             # probe: what the colour network's PyTorch-layer training costs (fine-tune: the feature maps are frozen too)
        surf.color_network.requires_grad_(False)
    graph = "--graph" in sys.argv               # the WHOLE step captured by the caller into one HIP graph (gens_amd.graph.GraphedStep): the lower bound
    no_auto = "--no-auto" in sys.argv           # every model call eager: what rounds 1 - 4 called the eager step
    foreach = "--foreach-adam" in sys.argv      # torch's multi-tensor Adam instead of the fused update get_optim_params asks for
    adam = {"capturable": True} if graph else {}
    finetune = "--finetune" in sys.argv          # BASELINE config 5 shape: volumes are the parameters, no volume build in the step
    full = "--full" in sys.argv                  # BASELINE config 3 as runner.py runs it: GenS.forward with the 2-D CNN (twice: the frozen matching copy
    #                                              too) and the 3-D U-Net inside the step
    saved_env = os.environ.get("GENS_FUSED_ADAM")
    if foreach:
        os.environ["GENS_FUSED_ADAM"] = "0"
    model = None
    if finetune:
        # GenS with its per-scene parameters in place (what runner.py:87-93 builds with init_volumes / load_params_vol), forward("train", ...) with
        # view_ids (runner.py:296-300)
        from gens_amd.models import gens
        torch.manual_seed(0)
        model = gens.GenS(gens_model_conf(volume_dims=tuple(dims), has_vol=True)).to(dev).train()
        with torch.no_grad():
            _, ft_masks = ops.volume_build([f.detach() for f in feats[:len(dims)]], intrs, c2ws, dims)
        model.volumes = torch.nn.ParameterList([torch.nn.Parameter(v.detach(), requires_grad=True) for v in vols])
        model.mask_volmes = torch.nn.ParameterList([torch.nn.Parameter(m, requires_grad=False) for m in ft_masks])
        model.features = torch.nn.ParameterList([torch.nn.Parameter(f.detach(), requires_grad=False) for f in feats])
        model._drop_captured_steps()
        surf = model.implicit_surface
        if "--freeze-color" in sys.argv:
            surf.color_network.requires_grad_(False)
        ipts["view_ids"] = list(range(nv))
        loss_fn = ft_loss
        opt = torch.optim.Adam(model.get_optim_params({"mlp_lr": 5e-4, "vol_lr": [5e-4] * len(dims)}), **adam)        # runner.py:96-97

        def forward():
            return model("train", ipts, cos_anneal_ratio=0.5), None
    elif full:
        from gens_amd.models import gens
        if os.environ.get("GENS_TRAIN_TINY"):        # probe: stand-in backbones ("feature", "reg" or "feature,reg")
            from tests.test_hip_ddp import TinyFeatureNet, TinyRegNet
            gens._BACKBONES.clear()
            which = os.environ["GENS_TRAIN_TINY"].split(",")
            gens.register_backbones(TinyFeatureNet if "feature" in which else None, TinyRegNet if "reg" in which else None)
        torch.manual_seed(0)
        model = gens.GenS(gens_model_conf(volume_dims=tuple(dims))).to(dev).train()
        surf = model.implicit_surface
        loss_fn = train_loss
        opt = torch.optim.Adam(model.get_optim_params({"mlp_lr": 5e-4, "feat_lr": 1e-3}), **adam)

        def forward():
            return model("train", ipts, cos_anneal_ratio=0.5, step=1), None
    else:
        loss_fn = train_loss
        opt = torch.optim.Adam([p for p in surf.parameters() if p.requires_grad], lr=5e-4, **({"fused": True} if not foreach else {}), **adam)

        def forward():
            cost, masks = ops.volume_build(feats[:len(dims)], intrs, c2ws, dims)        # K1 with autograd to the features, once per step (gens.py:139)
            out = surf("train", ipts, vols, masks, feats, [f.detach() for f in feats], 0.5, 1.0)
            return out, 1e-6 * sum(c.mean() for c in cost)                               # the cost volumes stand in for the U-Net's use of them
    if saved_env is None:
        os.environ.pop("GENS_FUSED_ADAM", None)
    else:
        os.environ["GENS_FUSED_ADAM"] = saved_env
    if no_auto or graph:
        (model if model is not None else surf).auto_graph = False

    def body():
        out, extra = forward()
        loss = loss_fn(out, targets)["loss"]                                             # runner.py:161-162 / 304-305
        if extra is not None:
            loss = loss + extra
        if os.environ.get("GENS_AG_SYNC"):
            torch.cuda.synchronize()
            sys.stderr.write("      [loop] loss computed\n")
        loss.backward()
        if os.environ.get("GENS_AG_SYNC"):
            torch.cuda.synchronize()
            sys.stderr.write("      [loop] backward returned\n")
        opt.step()
        if os.environ.get("GENS_AG_SYNC"):
            torch.cuda.synchronize()
            sys.stderr.write("      [loop] optimizer stepped\n")
        return loss.detach()

    def step():
        opt.zero_grad(set_to_none=True)
        if model is None:
            for t in feats + vols:               # leaves that stand in for the CNNs' outputs: a fresh gradient per step, as for a non-leaf
                t.grad = None
        return body()

    if graph:
        from gens_amd.graph import GraphedStep
        for t in feats + vols:                   # (leaves outside the optimiser: their gradients are assigned by every replay too)
            t.grad = None
        graphed = GraphedStep(body, [surf], opt, modules=[model] if full else ())

        def step():  # noqa: F811
            return graphed()

    trace = os.environ.get("GENS_TRAIN_TRACE")
    for i_ in range(int(sys.argv[sys.argv.index("--warm") + 1]) if "--warm" in sys.argv else 2):
        lv_ = float(step())
        if trace:
            sys.stderr.write("   loss %r\n" % lv_)
            torch.cuda.synchronize()
            sys.stderr.write("   warm-up step %d done\n" % i_)
            sys.stderr.flush()
    torch.cuda.synchronize()
    # What the process built so far (modules, plans, cached tensors -- and, inside bench.py, everything the earlier workloads left) goes out of the
    # cyclic collector's way, as bench.py does for the headline: a step creates tens of thousands of short-lived Python objects, i.e. a full
    # collection every few steps, and each of those walks every long-lived object of the process (~20 - 35 ms here).  Measured in bench.py's
    # process: the full step's median 52 - 54 ms with p10 32.4 before, against 32.8 ms in a process of its own.
    import gc
    if trace:
        graphs = [o for o in gc.get_objects() if isinstance(o, torch.cuda.CUDAGraph)]
        sys.stderr.write("   live CUDAGraph objects before the collection: %d\n" % len(graphs))
        del graphs
    n_coll = 0 if os.environ.get("GENS_TRAIN_NO_COLLECT") else gc.collect()
    if trace:
        graphs = [o for o in gc.get_objects() if isinstance(o, torch.cuda.CUDAGraph)]
        sys.stderr.write("   collected %d objects; live CUDAGraph objects after: %d\n" % (n_coll, len(graphs)))
        del graphs
    if not os.environ.get("GENS_TRAIN_NO_COLLECT"):
        gc.freeze()
    if trace:
        torch.cuda.synchronize()
        sys.stderr.write("   collected; timed loop starts\n")
        sys.stderr.flush()
        if trace == "2" and full:
            torch.cuda.memory._dump_snapshot("gpurun_out/snap_full.pickle")
    t0 = time.perf_counter()
    n = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 20
    host = 0.0
    per_step = []
    for _ in range(n):
        t1 = time.perf_counter()
        loss = step()
        host += time.perf_counter() - t1                                 # the host's share: every launch of the step enqueued
        float(loss)                                                      # (runner.py reads the loss every step: one synchronisation per step)
        surf.check_deferred()                                            # the reference's mid-step errors (already raised inside backward for a captured step)
        per_step.append(time.perf_counter() - t1)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    per_step.sort()
    _measure.stats = {"median_ms": round(per_step[len(per_step) // 2] * 1e3, 3), "p10_ms": round(per_step[len(per_step) // 10] * 1e3, 3),
                      "p90_ms": round(per_step[(9 * len(per_step)) // 10] * 1e3, 3), "steps": n}
    gc.unfreeze()
    if graph:
        graphed.check()
    label = ("full (CNNs + hot path)" if full else "fine-tune" if finetune else "train") + (", whole step as one HIP graph" if graph else ", every call eager" if no_auto else ", captured behind forward()")
    auto = getattr(model if model is not None else surf, "_auto", None)
    _measure.stats["auto_graph"] = dict(auto.stats) if auto is not None else None
    if not quiet:
        print(f"{label} step: {dt * 1e3:.1f} ms  ({512 * 128 / dt / 1e6:.2f} M ray-samples/s, 512 rays; host enqueue {host / n * 1e3:.1f} ms of it, incl. the waits inside the step); "
              f"median {_measure.stats['median_ms']:.2f} ms, p10 {_measure.stats['p10_ms']:.2f}, p90 {_measure.stats['p90_ms']:.2f}")
    if os.environ.get("GENS_TRAIN_OPS"):         # which torch operators make up the step's launches (torch.profiler over one step)
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
            float(step())
            torch.cuda.synchronize()
        rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.count > 0 and getattr(e, "device_time_total", 0) > 0]
        rows.sort(key=lambda e: -e.device_time_total)
        print(f"{'op':38s} {'count':>6s} {'gpu ms':>8s}  shapes")
        for e in rows[:int(os.environ["GENS_TRAIN_OPS"])]:
            print(f"{e.key[:38]:38s} {e.count:6d} {e.device_time_total / 1e3:8.3f}  {str(e.input_shapes)[:110]}")
    if kernels:
        from gens_amd import lib as L
        if graph:                                # (a replay launches nothing through the C ABI's host side)
            return dt * 1e3, label, {}
        owner = model if model is not None else surf
        was = getattr(owner, "auto_graph", True)
        owner.auto_graph = False                 # the kernel table: one extra step with every launch made from the host (the same kernels)
        try:
            L.profile_begin()
            step()
            return dt * 1e3, label, L.profile_end()
        finally:
            owner.auto_graph = was
    return dt * 1e3, label


if __name__ == "__main__":
    main()
