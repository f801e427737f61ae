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


def test_graphed_finetune_step_walks_the_eager_trajectory():
    from gens_amd.graph import GraphedStep
    warm, n = 2, 4
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
    step = GraphedStep(body_of(model2, opt2), [model2.implicit_surface], opt2, warmup=warm)
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
    opt = torch.optim.Adam(model.get_optim_params({"mlp_lr": 5e-4, "vol_lr": [1e-2, 1e-2, 1e-2]}))
    with pytest.raises(AssertionError, match="capturable"):
        GraphedStep(lambda: None, [model.implicit_surface], opt)
