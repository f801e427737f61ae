"""The captured training step BEHIND the reference's unchanged boundary (gens_amd.graph.AutoGraph inside GenS.forward / ImplicitSurface.forward).

The loops below are runner.py:157-166 / 300-308 as written -- `outputs = model(mode, inputs, ...)`, the caller's loss, `optimizer.zero_grad()`,
`loss.backward()`, `optimizer.step()`, the loss read back -- with a plain torch.optim.Adam and NO graph object in the caller.  After two eager calls
the model captures forward and backward into two HIP graphs and replays them; the run must walk the trajectory of the same loop with the
capture switched off (model.auto_graph = False), within the tolerances tests/test_hip_graph.py uses for GraphedStep."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _finetune_model(auto):
    from tests.test_hip_ddp import _inputs, _model
    model = _model()
    ipts = _inputs(7, nv=3)
    model.init_volumes({k: ipts[k] for k in ("imgs", "intrs", "c2ws")})
    model.auto_graph = auto
    return model


def _step_inputs(k, n_rays=48, nv=3):
    """A different ray batch, pseudo-point set and view order per step (what the fine-tune loop's dataset hands over, runner.py:296-297)."""
    from tests.test_hip_ddp import _inputs
    ipts = _inputs(100 + k, n_rays=n_rays, nv=nv)
    ipts["view_ids"] = [[0, 1, 2], [1, 0, 2], [2, 1, 0]][k % 3]
    ipts["color"] = torch.rand(n_rays, 3, generator=torch.Generator().manual_seed(k)).cuda()
    return ipts


def _runner_loop(model, opt, n, loss_fn, anneal=lambda k: 1.0, mode="train", step_of=lambda k: None, inputs=_step_inputs, zero=None):
    """runner.py:295-308 (fine-tune) / 154-166 (train): nothing here knows about graphs."""
    losses = []
    for k in range(n):
        ipts = inputs(k)
        outputs = model(mode, ipts, cos_anneal_ratio=anneal(k), step=step_of(k))
        loss = loss_fn(outputs, ipts)
        if zero is None:
            opt.zero_grad()
        else:
            opt.zero_grad(set_to_none=zero)
        loss.backward()
        opt.step()
        psnr = 20.0 * torch.log10(1.0 / (((outputs["color_fine"] - ipts["color"]) ** 2).mean()).sqrt())       # runner.py:311 reads the outputs after the update
        losses.append((float(loss), float(psnr)))
        del outputs
    torch.cuda.synchronize()
    return losses


def _compare(model_a, model_b, la, lb, rel=2e-5, prel=2e-4):
    assert len(la) == len(lb)
    for (a, pa), (b, pb) in zip(la, lb):
        assert abs(a - b) <= rel * abs(a), (la, lb)
        assert abs(pa - pb) <= 1e-3, (la, lb)
    assert len({a for a, _ in lb}) == len(lb)                 # different steps, not one step replayed
    pa = dict(model_a.named_parameters())
    for k, v in model_b.named_parameters():
        if v.requires_grad:
            ref = pa[k]
            assert float((v - ref).abs().max()) <= prel * max(float(ref.abs().max()), 1e-2), k      # (floor: parameters whose gradient is round-off, e.g. a bias in front of a softmax, move by Adam steps of noise)


def test_finetune_loop_is_captured_behind_the_unchanged_boundary():
    from tests.test_hip_ddp import _loss
    lrs = {"mlp_lr": 5e-4, "vol_lr": [1e-2, 1e-2, 1e-2]}
    n = 7
    runs = {}
    for auto in (False, True):
        model = _finetune_model(auto)
        opt = torch.optim.Adam(model.get_optim_params(lrs))       # runner.py:96-97, as written
        torch.manual_seed(21)
        runs[auto] = (model, _runner_loop(model, opt, n, _loss, anneal=lambda k: min(1.0, 0.2 * k)))
    model, graphed = runs[True]
    assert model._auto.stats["captured"] == 1 and model._auto.stats["replayed"] == n - 2 and model._auto.stats["superset_backward"] == 0, model._auto.stats
    assert not hasattr(model.implicit_surface, "_auto")           # the render did not capture a step of its own inside the model's
    _compare(runs[False][0], model, runs[False][1], graphed)


def test_captured_loop_follows_in_place_zeroing_and_gradient_accumulation():
    """optimizer.zero_grad(set_to_none=False) (torch 1.13's default, the version the reference pins) keeps .grad alive across steps: it must never
    alias a static gradient buffer that the next replay overwrites (the gradient would be added to itself)."""
    from tests.test_hip_ddp import _loss
    lrs = {"mlp_lr": 5e-4, "vol_lr": [1e-2, 1e-2, 1e-2]}
    runs = {}
    for auto in (False, True):
        model = _finetune_model(auto)
        opt = torch.optim.Adam(model.get_optim_params(lrs))
        torch.manual_seed(5)
        runs[auto] = (model, _runner_loop(model, opt, 6, _loss, zero=False))
    assert runs[True][0]._auto.stats["replayed"] == 4
    _compare(runs[False][0], runs[True][0], runs[False][1], runs[True][1])


def test_a_new_shape_is_a_new_capture_and_an_old_one_is_found_again():
    from tests.test_hip_ddp import _loss
    lrs = {"mlp_lr": 5e-4, "vol_lr": [1e-2, 1e-2, 1e-2]}
    sizes = [48, 48, 48, 48, 32, 32, 32, 32, 48, 32]
    runs = {}
    for auto in (False, True):
        model = _finetune_model(auto)
        opt = torch.optim.Adam(model.get_optim_params(lrs))
        torch.manual_seed(9)
        runs[auto] = (model, _runner_loop(model, opt, len(sizes), _loss, inputs=lambda k: _step_inputs(k, n_rays=sizes[k])))
    st = runs[True][0]._auto.stats
    assert st["captured"] == 2 and st["replayed"] == 2 + 2 + 2, st
    _compare(runs[False][0], runs[True][0], runs[False][1], runs[True][1])


def test_the_reference_errors_of_a_captured_step_are_raised_before_the_update():
    """No valid pseudo point (implicit_surface.py:494-495): the reference raises inside forward.  A captured step learns it from a device flag;
    it must surface inside loss.backward() -- before optimizer.step() -- without any help from the loop."""
    from tests.test_hip_ddp import _loss
    model = _finetune_model(True)
    opt = torch.optim.Adam(model.get_optim_params({"mlp_lr": 5e-4, "vol_lr": [1e-2, 1e-2, 1e-2]}))
    torch.manual_seed(3)
    _runner_loop(model, opt, 4, _loss)
    assert model._auto.stats["replayed"] == 2
    before = {k: v.detach().clone() for k, v in model.named_parameters()}
    ipts = _step_inputs(4)
    ipts["pseudo_pts"] = torch.full_like(ipts["pseudo_pts"], 5.0)          # far outside every mask volume
    outputs = model("train", ipts, cos_anneal_ratio=1.0)
    loss = _loss(outputs, ipts)
    opt.zero_grad()
    with pytest.raises(RuntimeError, match="No valid pseudo pts"):
        loss.backward()
    for k, v in model.named_parameters():
        assert torch.equal(v, before[k]), k
    # the loop goes on with good inputs
    more = _runner_loop(model, opt, 2, _loss, inputs=lambda k: _step_inputs(k + 5))
    assert all(torch.isfinite(torch.tensor(a)) for a, _ in more)


def test_a_loss_that_starts_to_use_another_output_is_served_by_the_superset_backward_then_recaptured():
    from tests.test_hip_ddp import _loss

    def loss_b(out, ipts):
        return _loss(out, ipts) + 0.3 * (out["normal"] ** 2).sum() + 0.1 * out["weight_sum"].sum()
    lrs = {"mlp_lr": 5e-4, "vol_lr": [1e-2, 1e-2, 1e-2]}
    runs = {}
    for auto in (False, True):
        model = _finetune_model(auto)
        opt = torch.optim.Adam(model.get_optim_params(lrs))
        torch.manual_seed(13)
        first = _runner_loop(model, opt, 4, _loss)
        second = _runner_loop(model, opt, 4, loss_b, inputs=lambda k: _step_inputs(k + 4))
        runs[auto] = (model, first + second)
    st = runs[True][0]._auto.stats
    assert st["superset_backward"] == 1 and st["captured"] == 2 and st["replayed"] == 2 + 4, st
    _compare(runs[False][0], runs[True][0], runs[False][1], runs[True][1])


def test_train_mode_with_the_cnns_is_captured_and_the_match_refresh_step_stays_eager():
    """GenS.forward("train") with the 2-D CNN, K1 and the U-Net inside the captured step; `step` as runner.py:157 passes it (epoch + batch / len):
    the step with step % 5 == 0 copies the feature network into its matching twin between the two CNN passes and is never captured; steps >= 5
    warp with the matching features (another launch sequence: another capture)."""
    from tests.test_hip_ddp import _loss, _model
    steps = [0.0, 0.125, 0.25, 0.375, 0.5, 0.625, 5.0, 5.125, 5.25, 5.375, 5.5]
    runs = {}
    for auto in (False, True):
        model = _model()
        model.auto_graph = auto
        opt = torch.optim.Adam(model.get_optim_params({"mlp_lr": 5e-4, "feat_lr": 1e-3}))
        torch.manual_seed(17)
        runs[auto] = (model, _runner_loop(model, opt, len(steps), _loss, anneal=lambda k: min(1.0, steps[k] / 2.0), step_of=lambda k: steps[k],
                                          inputs=lambda k: _step_inputs(k, nv=4) | {"view_ids": None}))
    st = runs[True][0]._auto.stats
    # steps[0] and steps[6] refresh the matching network (eager, outside the cache); 0.125, 0.25 warm up, 0.375 .. 0.625 replay; 5.125, 5.25 warm up, 5.375, 5.5 replay
    assert st["captured"] == 2 and st["replayed"] == 3 + 2, st
    _compare(runs[False][0], runs[True][0], runs[False][1], runs[True][1], rel=1e-4, prel=2e-3)


def test_render_alone_is_captured_when_it_is_called_directly():
    """ImplicitSurface.forward is a boundary of its own (implicit_surface.py:472-499): called directly with leaf volumes / feature maps the caller
    optimises (used where they are) and per-step mask volumes (copied), it captures its own step."""
    from gens_amd import ops
    from tests.test_hip_ddp import _loss
    runs = {}
    for auto in (False, True):
        model = _finetune_model(auto)
        surf = model.implicit_surface
        surf.auto_graph = auto
        vols = [v.detach().clone().requires_grad_(True) for v in model.volumes]
        feats = [f.detach().clone().requires_grad_(True) for f in model.features]
        masks = [m.detach().clone() for m in model.mask_volmes]
        opt = torch.optim.Adam(list(surf.parameters()) + vols + feats, lr=1e-3)
        torch.manual_seed(2)
        losses = []
        for k in range(6):
            ipts = _step_inputs(k)
            out = surf("train", ipts, vols, [m.clone() for m in masks], feats, feats, min(1.0, 0.3 * k), 1.0)
            loss = _loss(out, ipts)
            opt.zero_grad()
            loss.backward()
            opt.step()
            losses.append((float(loss), 0.0))
        runs[auto] = (surf, losses, vols, feats)
    surf = runs[True][0]
    assert surf._auto.stats["captured"] == 1 and surf._auto.stats["replayed"] == 4, surf._auto.stats
    _compare(runs[False][0], surf, runs[False][1], runs[True][1])
    for a, b in zip(runs[False][2] + runs[False][3], runs[True][2] + runs[True][3]):
        assert float((a - b).abs().max()) <= 2e-4 * max(float(a.abs().max()), 1e-3)
    assert ops is not None


def test_the_views_of_a_step_come_out_of_the_frozen_maps_and_their_layouts_in_one_launch():
    """GenS._select_frozen_views against torch.index_select on every frozen map and on the cached texel / warp layouts, bit for bit, for permuted
    subsets of the views and a repeated one; one gens_select_views launch, no index_select."""
    from gens_amd import lib as L, ops
    from gens_amd.ops.base import pack_maps
    model = _finetune_model(False)
    feats = list(model.features)
    nv = feats[0].shape[0]
    for ids in ([2, 0, 1][:nv], [nv - 1, 0], [1, 1, 0]):
        index = torch.tensor(ids, dtype=torch.long, device="cuda")
        L.profile_begin()
        got = model._select_frozen_views(index)
        names = [name for name, _, _, _ in L.profile_end(raw=True)]
        assert names.count("gens_select_views") == 1, names
        full = pack_maps(feats)
        for f, g, t in zip(feats, got, full):
            assert torch.equal(g, f.index_select(0, index))
            ver, _, tex = g._gens_tex
            assert ver == g._version and torch.equal(tex, t.index_select(0, index))
        if len(feats) >= 3:
            warp_full, channels = ops.build_warp_features(feats[:3])
            key, kept, (warp, ch) = got[0]._gens_warp
            assert ch == channels and torch.equal(warp, warp_full.index_select(0, index))
            assert key == tuple((id(f), f._version) for f in got[:3]) and all(a is b for a, b in zip(kept, got[1:3]))


def test_masks_that_arrive_with_their_bits_keep_them_through_a_captured_step():
    """The volume build hands out its masks with their bit-packed copies (ops.volume_build -> `_gens_bits`): the captured step holds no packing launch,
    every replay copies the step's words beside the step's masks -- or packs them when a step's masks come bare.  The masks CHANGE from step to step
    here (another visibility threshold), so stale words would show in the trajectory."""
    from gens_amd import ops
    from tests.test_hip_ddp import _loss
    runs = {}
    for auto in (False, True):
        model = _finetune_model(auto)
        surf = model.implicit_surface
        surf.auto_graph = auto
        vols = [v.detach().clone().requires_grad_(True) for v in model.volumes]
        feats = [f.detach().clone().requires_grad_(True) for f in model.features]
        dims = [int(v.shape[-1]) for v in vols]
        opt = torch.optim.Adam(list(surf.parameters()) + vols, lr=1e-3)
        torch.manual_seed(2)
        losses, seen = [], set()
        for k in range(7):
            ipts = _step_inputs(k)
            with torch.no_grad():
                _, masks = ops.volume_build([f.detach() for f in feats[:len(dims)]], ipts["intrs"], ipts["c2ws"], dims, min_vis_view=k % 3)
            assert all(hasattr(m, "_gens_bits") for m in masks)
            seen.add(tuple(int((m > 0).sum()) for m in masks))
            if k == 4:
                masks = [m.clone() for m in masks]             # a step whose masks come without their words
            out = surf("train", ipts, vols, masks, feats, feats, 0.5, 1.0)
            loss = _loss(out, ipts)
            opt.zero_grad()
            loss.backward()
            opt.step()
            losses.append((float(loss), 0.0))
        assert len(seen) == 3
        runs[auto] = (surf, losses, vols)
    surf = runs[True][0]
    assert surf._auto.stats["captured"] == 1 and surf._auto.stats["replayed"] == 5, surf._auto.stats
    entry = next(e for e in surf._auto.entries.values() if e.state == "captured")
    assert len(entry.mask_words) == len(runs[True][2])
    _compare(runs[False][0], surf, runs[False][1], runs[True][1])
    for a, b in zip(runs[False][2], runs[True][2]):
        assert float((a - b).abs().max()) <= 2e-4 * max(float(a.abs().max()), 1e-3)


@pytest.mark.parametrize("auto", [False, True])
def test_training_steps_leave_nothing_behind(auto):
    """Device memory and live autograd nodes after every step of the runner's loop: constant from the first step on (eager) / from the first
    replay on (captured).  Until round 5 every eager step leaked its `_SdfTrain` node -- ctx.sel held the Function's own outputs, a reference cycle
    through C++ that Python's collector cannot break -- and, through it, the blending node and everything both held on the device."""
    import gc

    from tests.test_hip_ddp import _loss
    model = _finetune_model(auto)
    opt = torch.optim.Adam(model.get_optim_params({"mlp_lr": 5e-4, "vol_lr": [1e-2, 1e-2, 1e-2]}))
    torch.manual_seed(4)
    seen = []
    for k in range(9):
        _runner_loop(model, opt, 1, _loss, inputs=lambda _k, k=k: _step_inputs(k))
        gc.collect()
        nodes = sum(1 for o in gc.get_objects() if type(o).__name__.endswith("Backward") and isinstance(o, torch.autograd.function.BackwardCFunction))
        seen.append((torch.cuda.memory_allocated(), nodes))
    settled = seen[4:]                                         # (captured: steps 0, 1 eager, step 2 captures; from then on replays)
    assert len({m for m, _ in settled}) == 1, seen           # bytes on the device
    assert len({n for _, n in settled}) == 1, seen           # custom autograd nodes alive (the last step's may still be referenced: not growing)


def test_two_models_trained_one_after_the_other_in_one_process():
    """bench.py trains one model after the other in ONE process; the second captured step used to die with a GPU memory fault on its second
    replay (a memset node inside K1's backward, profiles/r05_k1_bwd_graph_fault.txt).  Train mode (K1 forward + backward inside the captured
    graphs) after a fine-tune model, enough replays for the pool to have been reused."""
    from tests.test_hip_ddp import _loss, _model
    first = _finetune_model(True)
    opt = torch.optim.Adam(first.get_optim_params({"mlp_lr": 5e-4, "vol_lr": [1e-2, 1e-2, 1e-2]}))
    torch.manual_seed(8)
    _runner_loop(first, opt, 5, _loss)
    del first, opt
    second = _model()
    opt2 = torch.optim.Adam(second.get_optim_params({"mlp_lr": 5e-4, "feat_lr": 1e-3}))
    out = _runner_loop(second, opt2, 8, _loss, step_of=lambda k: 1.0 + k / 16, inputs=lambda k: _step_inputs(k, nv=4) | {"view_ids": None})
    assert second._auto.stats["replayed"] == 6
    assert all(torch.isfinite(torch.tensor(a)) for a, _ in out)


def test_capture_survives_another_thread_that_calls_the_runtime():
    """torch.distributed's RCCL threads (watchdog, proxy) call the runtime from their own threads whenever a collective is outstanding -- event queries,
    but also event creation / destruction and allocations -- and DDP broadcasts the buffers right before every forward.  In the default ("global")
    capture mode an allocation or an event call from ANY thread while a capture is open fails in that thread (which is how those threads abort the
    process) and invalidates the capture: a DDP-wrapped model died of it in one full run of this suite out of three.  The graphs are captured in
    thread-local mode: a thread that allocates, frees, creates and destroys events all the way through the loop changes nothing.
    (GENS_CAPTURE_MODE=global makes this test fail: the other thread's calls return hipErrorStreamCaptureUnsupported and the step stays eager.)"""
    import ctypes
    import threading
    from tests.test_hip_ddp import _loss
    hip = ctypes.CDLL("libamdhip64.so")
    stop, seen = threading.Event(), []

    def other_thread():
        try:
            torch.cuda.set_device(0)
            n, bad = 0, 0
            while not stop.is_set():
                ptr, ev = ctypes.c_void_p(), ctypes.c_void_p()
                rc = [hip.hipMalloc(ctypes.byref(ptr), ctypes.c_size_t(1 << 16)), hip.hipEventCreate(ctypes.byref(ev))]
                if rc[1] == 0:
                    rc += [hip.hipEventQuery(ev) if False else 0, hip.hipEventDestroy(ev)]
                if rc[0] == 0:
                    rc.append(hip.hipFree(ptr))
                bad += sum(1 for r in rc if r != 0)
                n += 1
            seen.append((n, bad))
        except Exception as e:  # noqa: BLE001
            seen.append(e)

    model = _finetune_model(True)
    opt = torch.optim.Adam(model.get_optim_params({"mlp_lr": 5e-4, "vol_lr": [1e-2, 1e-2, 1e-2]}))
    th = threading.Thread(target=other_thread, daemon=True)
    th.start()
    try:
        losses = _runner_loop(model, opt, 6, _loss)
    finally:
        stop.set()
        th.join(30)
    assert len(seen) == 1 and isinstance(seen[0], tuple), seen
    n, bad = seen[0]
    assert n > 100 and bad == 0, seen                                                    # the other thread ran all the way and no call of its failed
    stats = model._auto.stats
    assert stats["captured"] == 1 and stats["replayed"] == 4 and stats["eager"] == 2, stats
    assert all(torch.isfinite(torch.tensor(l)).all() for l in losses)


def test_a_capture_that_fails_leaves_the_loop_on_the_eager_path_of_the_same_process():
    """Something inside the step that a capture cannot hold (here: a stream synchronisation the first time the step is captured) must cost a warning,
    not the run: the signature stays eager, the loop goes on in this process and walks the un-captured trajectory."""
    import warnings
    from tests.test_hip_ddp import _loss
    lrs = {"mlp_lr": 5e-4, "vol_lr": [1e-2, 1e-2, 1e-2]}
    runs = {}
    for auto in (False, True):
        model = _finetune_model(auto)
        opt = torch.optim.Adam(model.get_optim_params(lrs))
        if auto:
            inner, calls = model._forward_impl, []

            def spoiled(*a, **k):
                calls.append(torch.cuda.is_current_stream_capturing())
                if calls[-1]:
                    torch.cuda.current_stream().synchronize()           # illegal while capturing
                return inner(*a, **k)
            model._forward_impl = spoiled
        torch.manual_seed(21)
        with warnings.catch_warnings(record=True) as seen:
            warnings.simplefilter("always")
            runs[auto] = (model, _runner_loop(model, opt, 6, _loss))
        if auto:
            assert any("cannot be captured" in str(w.message) for w in seen), [str(w.message) for w in seen]
            assert calls.count(True) == 1, calls                          # one capture attempt, then eager for good
            stats = model._auto.stats
            assert stats["captured"] == 0 and stats["replayed"] == 0 and stats["eager"] == 6, stats
    _compare(runs[False][0], runs[True][0], runs[False][1], runs[True][1])
    assert torch.isfinite(torch.randn(8, device="cuda")).all()        # the device's generator is out of capture mode again (the next test's randn raised once)


def test_frozen_feature_layouts_are_taken_from_the_cached_pyramid_bit_for_bit():
    """GenS(has_vol): the texel / warp layouts of this step's views, index-selected out of the layouts of the whole frozen pyramid (built once), against the
    layouts packed from `features[i][view_ids]` every step (the per-tensor cache switched off): the same outputs bit for bit, for several view orders, and
    after the frozen maps change in place (their version moves: the cached layouts are rebuilt)."""
    from gens_amd.ops.base import kernels
    from tests.test_hip_ddp import _inputs
    model = _finetune_model(False)
    f0 = model.features[0].detach().clone()
    runs = {}
    for cache in (True, False):
        kernels.tex_cache = cache
        try:
            outs = []
            for k, ids in enumerate(([0, 1, 2], [2, 0, 1], [1, 2, 0], [0, 1, 2])):
                if k == 3:
                    with torch.no_grad():
                        model.features[0].mul_(1.5)                       # frozen, not constant: e.g. a checkpoint loaded in place
                ipts = _inputs(300 + k, n_rays=32, nv=3)
                ipts["view_ids"] = ids
                torch.manual_seed(5 + k)
                with torch.no_grad():
                    out = model("train", ipts, cos_anneal_ratio=0.5, step=None)
                outs.append({n: v.clone() for n, v in out.items() if torch.is_tensor(v)})
            runs[cache] = outs
            with torch.no_grad():
                model.features[0].copy_(f0)
        finally:
            kernels.tex_cache = True
    for a, b in zip(runs[True], runs[False]):
        assert sorted(a) == sorted(b)
        for n in a:
            assert torch.equal(a[n], b[n]), n
    assert not torch.equal(runs[True][0]["color_fine"], runs[True][3]["color_fine"])          # (the in-place change reached the render)
