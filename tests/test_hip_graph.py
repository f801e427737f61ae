"""gens_amd.graph.GraphedStep: a fine-tune step (GenS.forward + loss + backward + Adam) captured once into a HIP graph and replayed must walk the
same trajectory as the eager loop -- same generator draws, same losses, same parameters."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup():
    from tests.test_hip_ddp import _inputs, _loss, _model
    model = _model()
    ipts = _inputs(7, nv=3)
    model.init_volumes({k: ipts[k] for k in ("imgs", "intrs", "c2ws")})
    ipts["view_ids"] = [0, 1, 2]
    lrs = {"mlp_lr": 5e-4, "vol_lr": [1e-2, 1e-2, 1e-2]}
    opt = torch.optim.Adam(model.get_optim_params(lrs), capturable=True)
    return model, ipts, opt, _loss


@pytest.mark.parametrize("warm", [2, None])
def test_graphed_finetune_step_walks_the_eager_trajectory(warm):
    """warm=None: GraphedStep's own default (3 warm-up calls).  The model's own capture-behind-the-boundary (AutoGraph, on by default) would start
    on exactly that third call, nested inside this warm-up: GraphedStep switches it off for its warm-up and its capture."""
    from gens_amd.graph import GraphedStep
    n = 4
    # eager: n steps from the seed.  The graphed run warms up `warm` times and captures once in between -- and must STILL walk this trajectory:
    # GraphedStep puts parameters, optimiser state and the CPU generator back, so replay k is step k
    model, ipts, opt, loss_fn = _setup()
    surf = model.implicit_surface

    def body_of(model, opt):
        def body():
            loss = loss_fn(model("finetune", ipts, cos_anneal_ratio=1.0, step=None))
            loss.backward()
            opt.step()
            return loss.detach()
        return body

    torch.manual_seed(21)
    eager, body = [], body_of(model, opt)
    for i in range(n):
        opt.zero_grad(set_to_none=True)
        eager.append(float(body()))
    surf.check_deferred()
    eager_params = {k: v.detach().clone() for k, v in model.named_parameters() if v.requires_grad}

    model2, ipts2, opt2, _ = _setup()
    for (k, a), (_, b) in zip(model.state_dict().items(), model2.state_dict().items()):
        assert a.shape == b.shape
    torch.manual_seed(21)
    ipts = ipts2                                     # (body_of closes over `ipts`: the second model's own tensors)
    step = GraphedStep(body_of(model2, opt2), [model2.implicit_surface], opt2, **({} if warm is None else {"warmup": warm}))
    assert getattr(model2, "_auto", None) is None or model2._auto.stats["captured"] == 0      # no capture of the model's own inside the caller's
    graphed = []
    for _ in range(n):
        graphed.append(float(step()))
        step.check()
    assert len(eager) == len(graphed) == n
    for a, b in zip(eager, graphed):
        assert abs(a - b) <= 2e-5 * abs(a), (eager, graphed)
    assert len(set(graphed)) == n                    # the replays are different steps (new draws, new weights), not one step n times
    for k, v in model2.named_parameters():
        if v.requires_grad:
            ref = eager_params[k]
            assert float((v - ref).abs().max()) <= 2e-4 * max(float(ref.abs().max()), 1e-3), k
    # a loop that never reads the loss back: every replay waits for the previous one's copy of the draw buffer (no torn / duplicated draws)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    step.check()
    assert torch.isfinite(step.loss).all()


def test_graphed_step_wants_a_capturable_optimiser():
    from gens_amd.graph import GraphedStep
    model, ipts, _, loss_fn = _setup()
    groups = model.get_optim_params({"mlp_lr": 5e-4, "vol_lr": [1e-2, 1e-2, 1e-2]})
    for g in groups:
        g.pop("fused", None)                                 # (get_optim_params asks for the fused update, which IS capturable: the multi-tensor default is not)
    opt = torch.optim.Adam(groups)
    with pytest.raises(AssertionError, match="capturable"):
        GraphedStep(lambda: None, [model.implicit_surface], opt)


def test_graphed_full_training_step_with_the_cnns_walks_the_eager_trajectory():
    """The WHOLE GenS.forward training step -- MnasNet trunk (MIOpen's dense convolutions, K21, K22) and its frozen matching copy, K1, the 3-D U-Net
    (K15 / K16), render, loss, backward, Adam -- captured and replayed: same losses, same parameters, and the BatchNorm running statistics and
    batch counters of n steps (not of n + warm-up)."""
    from gens_amd import synthetic
    from gens_amd.config import gens_model_conf
    from gens_amd.graph import GraphedStep
    from gens_amd.models import gens
    from tests.test_hip_ddp import _loss
    saved = dict(gens._BACKBONES)
    gens._BACKBONES.clear()                                  # this package's FeatureNetwork / RegNetwork, not another test's stand-ins
    try:
        def build():
            torch.manual_seed(0)
            model = gens.GenS(gens_model_conf(volume_dims=(32, 16, 8))).cuda().train()
            opt = torch.optim.Adam(model.get_optim_params({"mlp_lr": 5e-4, "feat_lr": 1e-3}), capturable=True)
            return model, opt
        sc = synthetic.make_scene(nv=4, h=64, w=96, n_levels=1, seed=5)
        g = torch.Generator().manual_seed(3)
        pix = torch.stack([torch.randint(4, 92, (48,), generator=g), torch.randint(4, 60, (48,), generator=g)], -1)
        ro, rd = synthetic.make_rays(sc["intrs"], sc["c2ws"], 64, 96, pixels=pix)
        ipts = {k: v.cuda() for k, v in {"imgs": sc["imgs"], "intrs": sc["intrs"], "c2ws": sc["c2ws"], "rays_o": ro, "rays_d": rd, "near": sc["near"],
                                         "far": sc["far"], "pseudo_pts": torch.rand(256, 3, generator=g) - 0.5}.items()}

        def body_of(model, opt):
            def body():
                loss = _loss(model("train", ipts, cos_anneal_ratio=0.5, step=1.0))
                loss.backward()
                opt.step()
                return loss.detach()
            return body

        n = 3

        def run_eager():
            model, opt = build()
            torch.manual_seed(33)
            losses, body = [], body_of(model, opt)
            for _ in range(n):
                opt.zero_grad(set_to_none=True)
                losses.append(float(body()))
            model.implicit_surface.check_deferred()
            return model, losses

        def distance(ma, mb):
            """largest parameter difference in units of what n Adam steps can move a parameter (n * lr, plus 1e-3 of its magnitude)"""
            ra, worst = dict(ma.named_parameters()), (0.0, None)
            for k, v in mb.named_parameters():
                if v.requires_grad:
                    d = float((v.detach() - ra[k].detach()).abs().max()) / (n * 1e-3 + 1e-3 * float(ra[k].abs().max()))
                    worst = max(worst, (d, k))
            return worst

        model, eager = run_eager()
        model_b, eager_b = run_eager()                       # the yardstick: two EAGER runs differ by the order of the float atomics behind the gradients,
        yard = distance(model, model_b)                      # which Adam turns into a few per cent of a step where a gradient nearly cancels (BatchNorm shifts,
        model2, opt2 = build()                               # weights in front of a normalisation)
        torch.manual_seed(33)
        step = GraphedStep(body_of(model2, opt2), [model2.implicit_surface], opt2, warmup=2, modules=[model2])
        graphed = []
        for _ in range(n):
            graphed.append(float(step()))
            step.check()
        for a, b in zip(eager, graphed):
            assert abs(a - b) <= 1e-4 * abs(a), (eager, graphed)
        assert len(set(graphed)) == n
        got = distance(model, model2)
        assert got[0] <= max(3.0 * yard[0], 0.02), (got, yard)
        refb = dict(model.named_buffers())
        counters = 0
        for k, b in model2.named_buffers():
            if k.endswith("num_batches_tracked"):
                counters += 1
                assert int(b) == int(refb[k]) == (n if "match_feature_network" not in k or int(refb[k]) else 0), (k, int(b), int(refb[k]))
            elif b.dtype.is_floating_point:
                # (running statistics after n steps of momentum 3e-4: n * 3e-4 of a batch statistic that itself carries the step's noise)
                assert float((b - refb[k]).abs().max()) <= 1e-3 * float(refb[k].abs().max()) + 1e-5, k
        assert counters > 50                                 # the trunk's BatchNorm layers were in the comparison
    finally:
        gens._BACKBONES.clear()
        gens._BACKBONES.update(saved)
