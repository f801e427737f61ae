"""Mirror of the reference's models/modules/implicit_surface.py (NeuS-style renderer), re-organised around the
per-ray HIP kernels of libgens_hip.so:

    reference method (file:line)              here
    ----------------------------------------  -----------------------------------------------------------------
    sample_pdf            :14-44              fused into ops.upsample (K6)
    up_sample             :60-109             ImplicitSurface.up_sample      -> ops.upsample (K5+K6, one wave per ray)
    cat_z_vals            :111-133            ImplicitSurface.cat_z_vals     -> ops.merge_samples (K7)
    tv_regularization     :135-150            ops.tv_regularization (K10)
    render_core           :152-349            ImplicitSurface.render_core    -> K3, K2(+K2''), K4, K8, K9
    render                :351-405            ImplicitSurface.render
    extract_geometry      :407-427            ImplicitSurface.extract_geometry (device lattice, one D2H copy)
    validate              :429-470            ImplicitSurface.validate
    forward               :472-499            ImplicitSurface.forward

Same constructor, method names, argument order and output keys as the reference, so models/gens.py drives it
unchanged.  Host RNG draws (`torch.rand([B,1])`, `torch.rand([1024,3])` from the CPU generator, :256,:362) are
kept in the reference's order so a seeded run renders the same jitter.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import ops
from .blending_network import BlendingNetwork
from .projector import lookup_feature, surface_patch_warp
from .sdf_network import SDFNetwork
from .variance_network import SingleVarianceNetwork

REFERENCE_CHUNK = 256      # rays per render() call in the reference's validate (:437-438)
N_RANDOM_PTS = 1024        # sparse-SDF probe points per render_core call (:256)


class Scene:
    """Per-scene device state shared by every ray chunk: mask pyramid, packed volumes, texel views, warp features."""

    def __init__(self, volumes, mask_volumes, imgs, features, match_features, intrs, c2ws):
        self.volumes = list(volumes)
        self.mask_volumes = list(mask_volumes)
        self.masks = ops.VolumeSet.masks(self.mask_volumes)
        self.imgs, self.features, self.match_features = imgs, features, match_features
        self.intrs, self.c2ws = intrs, c2ws
        self.views = ops.SceneViews(imgs, intrs, c2ws, features)
        self._packed = None
        self._warp = {}

    def ref_rotation(self):
        """Row-major inverse of the reference camera's rotation (implicit_surface.py:242,245): nine floats ON THE DEVICE from the scene's
        set-up launch (the compositing kernel reads them there; round 2 read them back to the host once per scene, a synchronisation per
        training step)."""
        return self.views.cams.rot_inv

    def volumes_nograd(self):
        """Packed (X,Y,Z,4) texel copy for passes that never need d/dvolume (sampling rounds, inference)."""
        if self._packed is None:
            self._packed = ops.VolumeSet.packed(self.volumes)
        return self._packed

    def warp_features(self, use_match):
        """cat([f0, up(f1), up(f2)]) texels, built once per scene (the reference rebuilds them per chunk, :313-326)."""
        if use_match not in self._warp:
            src = self.match_features if use_match else self.features
            self._warp[use_match] = ops.build_warp_features(src[:3])
        return self._warp[use_match]


def _jitter_block(n_rays, chunk=REFERENCE_CHUNK, generator=None):
    """Jitter of `n_rays` rays (a whole number of reference chunks, except possibly the last) in the reference's draw order:
    per chunk `torch.rand([chunk, 1])` followed by the `torch.rand([1024, 3])` of render_core (implicit_surface.py:256,362).
    The CPU generator produces one float per 32-bit draw, so ONE large call yields the same stream as the 2*k small ones.
    generator: None = the default CPU generator (the reference's), else a private one standing at the state the draws start from."""
    full, rem = divmod(n_rays, chunk)
    out = []
    if full:
        per = chunk + 3 * N_RANDOM_PTS
        out.append(torch.rand(full * per, generator=generator).reshape(full, per)[:, :chunk].reshape(-1, 1))
    if rem:
        out.append(torch.rand([rem, 1], generator=generator))
        torch.rand([N_RANDOM_PTS, 3], generator=generator)
    return torch.cat(out, 0) if out else torch.zeros(0, 1)


def reference_jitter(n_rays, chunk=REFERENCE_CHUNK):
    """The (n_rays,1) stratified-jitter draws `validate` would make chunk by chunk, so any chunking here renders the same image."""
    return _jitter_block(n_rays, chunk)


class JitterStream:
    """reference_jitter() produced group by group on a helper thread, so the ~4 M host RNG draws of a 480x640 image overlap
    with GPU work instead of preceding it.

    generator=None draws from the default CPU generator -- then nothing else may draw from it while the thread runs (validate() joins the
    thread before it returns).  With a PRIVATE generator (validate()'s head start on the next image, ImplicitSurface._speculate_jitter) the
    thread touches nothing shared: the draws become the default generator's past only if they are used (its state is moved to
    `end_state()` then)."""

    def __init__(self, n_rays, group, buf=None, generator=None):
        """buf: optional (n_rays, 1) host buffer to fill -- a page-locked one makes the per-chunk upload asynchronous."""
        import threading
        group = max(REFERENCE_CHUNK, group // REFERENCE_CHUNK * REFERENCE_CHUNK)
        self.bounds = [(s, min(s + group, n_rays)) for s in range(0, n_rays, group)]
        self.buf = buf if buf is not None else torch.empty(n_rays, 1)
        self.generator = generator
        self.ready = [threading.Event() for _ in self.bounds]
        self.thread = threading.Thread(target=self._run, daemon=False)    # (joined at interpreter exit: a daemon thread killed inside torch's generator aborts the process)
        self.thread.start()

    def _run(self):
        for k, (s, e) in enumerate(self.bounds):
            self.buf[s:e] = _jitter_block(e - s, generator=self.generator)
            self.ready[k].set()

    def slice(self, s, e):
        for k, (bs, be) in enumerate(self.bounds):
            if be <= s or bs >= e:
                continue
            self.ready[k].wait()
        return self.buf[s:e]

    def join(self):
        self.thread.join()

    def end_state(self):
        """(after join) the generator state behind the image's last draw"""
        return self.generator.get_state()


class ImplicitSurface(nn.Module):
    def __init__(self, confs):
        super().__init__()
        self.n_samples = confs.get_int("render.n_samples")
        self.n_importance = confs.get_int("render.n_importance")
        self.up_sample_steps = confs.get_int("render.up_sample_steps")
        self.perturb = confs.get_float("render.perturb")
        self.sdf_network = SDFNetwork(**confs["sdf_network"])
        self.color_network = BlendingNetwork(**confs["color_network"])
        self.deviation_network = SingleVarianceNetwork(**confs["variance_network"])
        self.val_chunk = None          # rays per chunk in validate(); rays are independent, so this is a free knob.  None: equal chunks of at most
                                       # 32 768 rays over the ray range (chunking.balanced_chunk) -- the setting every committed measurement uses
        self.speculate_jitter = True   # validate() draws the NEXT image's jitter ahead of time on a private generator (see _speculate_jitter)
        self.fused_sdf = True          # inference: evaluate the SDF network with the fused MFMA kernel (gens_sdf_mlp)
        self.sdf_precision = "f32"     # "f32": exact float32 MFMA; "f16x2": split-half operands (~1e-6 rel.), float32 fallback on overflow
        self._sdf_plan = None
        self.fused_blend = True        # inference: source-view look-up + colour network in one kernel (gens_blend_views)
        self._blend_plan = None
        self.fused_train = True        # training: SDF value / gradient / smooth and their backward in the K17 kernels (gens_sdf_train_*)
        self.fused_sampling = True     # one launch per sampling round (gens_merge_upsample) instead of merge + up-sample (+ mid-points) launches

    # ----------------------------------------------------------------------------------------------------------
    # masked SDF evaluation (Q7, Q8)
    # ----------------------------------------------------------------------------------------------------------
    @staticmethod
    def _select(valid):
        """Indices of valid points; if none is valid the first 10 are used (:123-124,176-177,372-373)."""
        idx = torch.nonzero(valid, as_tuple=False)[:, 0]
        if idx.numel() < 1:
            idx = torch.arange(min(10, valid.numel()), device=valid.device)
        return idx

    def _fused_plan(self, volumes):
        """The packed-weight plan for gens_sdf_mlp, or None when the fused kernel does not apply (autograd needed,
        planar volumes, or a non-shipped architecture) -- then the PyTorch layers run on top of the K2 kernels."""
        if not self.fused_sdf or torch.is_grad_enabled():
            return None
        if not (isinstance(volumes, ops.VolumeSet) and volumes.layout == 1 and 1 <= volumes.n <= 5):
            return None
        net = self.sdf_network
        if not ops.SdfMlpPlan.supported(net) or net.init_feat_channels != 4 * volumes.n:
            return None
        if self._sdf_plan is None or self._sdf_plan.key != ops.SdfMlpPlan.version(net):
            self._sdf_plan = ops.SdfMlpPlan(net)
        return self._sdf_plan

    def _fused_blend_plan(self, views, features=None, imgs=None):
        """The packed-weight plan of gens_blend_views, or None when the fused kernel does not apply.  It has no backward: under autograd it
        runs only if NOTHING on the colour path asks for a gradient (a frozen colour network over frozen feature maps / images)."""
        if not self.fused_blend:
            return None
        if torch.is_grad_enabled():
            wants = any(p.requires_grad for p in self.color_network.parameters())
            wants = wants or any(t.requires_grad for t in (features or [])) or (imgs is not None and imgs.requires_grad)
            if wants or features is None:
                return None
        net = self.color_network
        if not ops.BlendPlan.supported(net) or len(views.feat_tex) > 5 or net.rgb_fc[0].weight.shape[1] != 37:
            return None
        if self._blend_plan is None or self._blend_plan.key != ops.BlendPlan.version(net):
            self._blend_plan = ops.BlendPlan(net)
        return self._blend_plan if self._blend_plan.n_feat == 3 + 4 * len(views.feat_tex) else None

    def _precision(self, plan, want_grad=False):
        """The arithmetic of one SDF launch.  The split-half kernels pre-scale their weight streams by 100 / ln 2, so they have their own
        range condition (plan.value_ok / plan.grad_pieces: |w|, |b| below ~416); a network that fails it is evaluated in float32, it does
        not raise."""
        if self.sdf_precision != "f16x2":
            return "f32"
        if want_grad:
            return "f16x2" if getattr(plan, "grad_pieces", None) is not None else "f32"
        return "f16x2" if getattr(plan, "value_ok", False) else "f32"

    def _train_net(self, scene, lean=False):
        """This step's fused SDF evaluator (ops.SdfTrainStep: the effective weights packed once for the sampling passes, render_core
        and the backward), or None outside training / for architectures the K17 kernels do not cover."""
        if lean or not self.fused_train or not torch.is_grad_enabled():
            return None
        if not any(p.requires_grad for p in self.sdf_network.parameters()) and not any(v.requires_grad for v in scene.volumes):
            return None
        cached = getattr(scene, "_train_net_cache", None)        # one evaluator (weight norm + stream packing) per scene object = per step
        if cached is None:
            cached = scene._train_net_cache = (self.sdf_network.train_step(scene.volumes, scene.volumes_nograd(), tv_masks=scene.mask_volumes),)
        return cached[0]

    def _split_half_overflowed(self):
        """True if a split-half launch met a value outside the half range since the last check (one device sync)."""
        return self.sdf_precision == "f16x2" and self._sdf_plan is not None and self._sdf_plan.overflowed()

    def _masked_sdf(self, pts, valid, volumes, net=None):
        plan = net if net is not None else self._fused_plan(volumes)
        if plan is not None:              # compaction + count stay on the device: no host synchronisation; the unselected rows' 100 rides on it
            sdf = torch.empty(pts.shape[0], 1, device=pts.device, dtype=torch.float32)
            idx, count = ops.compact_fill(valid, sdf=sdf)
            ops.sdf_mlp(plan, volumes, pts, index=idx, sdf_out=sdf, precision=self._precision(plan), count=count)
        else:
            sdf = torch.full((pts.shape[0], 1), 100.0, device=pts.device, dtype=pts.dtype)
            idx = self._select(valid)
            sdf[idx] = self.sdf_network.sdf(pts[idx], volumes)
        return sdf

    # ----------------------------------------------------------------------------------------------------------
    # hierarchical sampling
    # ----------------------------------------------------------------------------------------------------------
    def up_sample(self, rays_o, rays_d, z_vals, sdf, n_importance, mask_volumes, inv_s):
        """Importance samples for a fixed inv_s (:60-109) -> (B, n_importance)."""
        return ops.upsample(rays_o, rays_d, z_vals, sdf.reshape(z_vals.shape), n_importance, mask_volumes, inv_s)[0]

    def cat_z_vals(self, rays_o, rays_d, z_vals, new_z_vals, sdf, volumes, mask_volumes, last=False):
        """Merge new depths (and, unless `last`, their SDF values) into the sorted sample list (:111-133)."""
        if last:
            return ops.merge_samples(z_vals, new_z_vals)[0], sdf
        pts, valid = ops.ray_points(rays_o, rays_d, new_z_vals, mask_volumes)
        new_sdf = self._masked_sdf(pts, valid, volumes).reshape(new_z_vals.shape)
        return ops.merge_samples(z_vals, new_z_vals, sdf.reshape(z_vals.shape), new_sdf)

    def tv_regularization(self, volume_feat_cas, volume_mask_cas=None):
        if volume_mask_cas is None:
            volume_mask_cas = [torch.ones_like(v[:, :1]) for v in volume_feat_cas]
        return ops.tv_regularization(list(volume_feat_cas), list(volume_mask_cas))

    @torch.no_grad()
    def _sample_rays(self, rays_o, rays_d, z_vals, scene, net=None, mid_points=None):
        """The hierarchical sampling of render() (:364-393).  One launch per round between two evaluations of the network: the merge of round i
        (cat_z_vals, :111-133) and the up-sampling of round i + 1 (:60-109) are one kernel (gens_merge_upsample); with `mid_points` =
        sample_dist the last merge also produces render_core's section mid-points and their mask decisions (:163-173), left in
        self._mid_points for the render_core call that follows with the returned z_vals.  `fused_sampling = False` runs the operators one by
        one (gens_upsample / gens_merge_samples: the same bits)."""
        masks, vols = scene.masks, scene.volumes_nograd()
        b = rays_o.shape[0]
        self._mid_points = None
        pts, valid = ops.ray_points(rays_o, rays_d, z_vals, masks)
        sdf = self._masked_sdf(pts, valid, vols, net).reshape(b, -1)
        n_new = self.n_importance // self.up_sample_steps
        valid = valid.reshape(b, -1)                       # mask decisions travel with the samples through the merges
        if not getattr(self, "fused_sampling", True) or self.up_sample_steps < 1:
            for i in range(self.up_sample_steps):
                z_new, pts_new, valid_new = ops.upsample(rays_o, rays_d, z_vals, sdf, n_new, masks, 64 * 2 ** i, valid_in=valid)
                if i + 1 == self.up_sample_steps:
                    z_vals, _ = ops.merge_samples(z_vals, z_new)
                else:
                    sdf_new = self._masked_sdf(pts_new, valid_new, vols, net).reshape(b, n_new)
                    z_vals, sdf, valid = ops.merge_samples(z_vals, z_new, sdf, sdf_new, valid, valid_new)
            return z_vals
        z_new, pts_new, valid_new = ops.upsample(rays_o, rays_d, z_vals, sdf, n_new, masks, 64, valid_in=valid)
        for i in range(1, self.up_sample_steps):
            sdf_new = self._masked_sdf(pts_new, valid_new, vols, net).reshape(b, n_new)
            z_vals, sdf, valid, z_new, pts_new, valid_new = ops.merge_upsample(rays_o, rays_d, z_vals, sdf, valid, z_new, sdf_new, valid_new, n_new, masks,
                                                                               64 * 2 ** i)
        if mid_points is None:
            z_vals, _ = ops.merge_samples(z_vals, z_new)
        else:
            z_vals, pts_mid, valid_mid = ops.merge_mid_points(rays_o, rays_d, z_vals, z_new, masks, mid_points)
            self._mid_points = (z_vals, float(mid_points), pts_mid, valid_mid)
        return z_vals

    # ----------------------------------------------------------------------------------------------------------
    # render_core
    # ----------------------------------------------------------------------------------------------------------
    def _host_draws(self, dev, b=None):
        """The host-generator draws of a render in the reference's order -- torch.rand([B, 1]) (:362, when b is given), then
        torch.rand([1024, 3]) (:256) -- staged through ONE page-locked buffer: a pageable host-to-device copy waits for the stream to drain
        on ROCm (the launch queue ran empty once per step), a pinned one is just another asynchronous launch.  -> (t_rand (B,1) or None,
        pts_random (1024,3) in [-1, 1)) on the device.  The ` * 2 - 1` of :256 happens on the host: the same two float32 operations."""
        n_t = 0 if b is None else b
        capturing = torch.cuda.is_current_stream_capturing()
        buf = getattr(self, "_pinned_draws", None)
        if buf is None or buf.numel() != n_t + 3 * N_RANDOM_PTS:
            buf = torch.empty(n_t + 3 * N_RANDOM_PTS, dtype=torch.float32, pin_memory=True)
            self._pinned_draws = buf
            self._pinned_draws_event = None
        elif self._pinned_draws_event is not None and not capturing:
            self._pinned_draws_event.synchronize()           # the previous step's copy out of this buffer has long finished
        self._pinned_draws_layout = (n_t, b)
        self.refresh_host_draws()
        on_dev = buf.to(dev, non_blocking=True)              # (captured into a graph this is a copy node that reads the buffer at every replay)
        if not capturing:
            self._pinned_draws_event = torch.cuda.Event()
            self._pinned_draws_event.record()
        return (on_dev[:n_t].view(b, 1) if n_t else None), on_dev[n_t:].view(N_RANDOM_PTS, 3)

    def refresh_host_draws(self, buf=None, layout=None):
        """Draw the step's host random numbers into the page-locked staging buffer (the reference's generator, its order and shapes).  A step
        replayed from a captured graph (gens_amd.graph.GraphedStep / AutoGraph) calls this before every replay: the graph's copy node then
        carries the new draws to the device.  The previous replay must have finished reading the buffer.  buf / layout: the buffer a captured
        step owns (AutoGraph keeps one per captured signature); default: this module's own."""
        n_t, b = self._pinned_draws_layout if layout is None else layout
        buf = self._pinned_draws if buf is None else buf
        if n_t:
            buf[:n_t] = torch.rand([b, 1]).reshape(-1)
        buf[n_t:] = (torch.rand([N_RANDOM_PTS, 3]) * 2 - 1).reshape(-1)

    def begin_capture(self):
        """Before a step is captured into a HIP graph: fresh page-locked buffers for the host draws and the deferred checks, allocated NOW (a
        page-locked allocation is not a capturable operation) in the sizes the eager calls before this one used; the captured copy nodes will
        read / write these, and end_capture() hands them to the graph's owner."""
        old = getattr(self, "_pinned_draws", None)
        self._pinned_draws = None if old is None else torch.empty(old.numel(), dtype=torch.float32, pin_memory=True)
        self._pinned_draws_event = None
        self._deferred_host = torch.zeros(4, dtype=torch.int32, pin_memory=True)
        self._deferred = None

    def end_capture(self):
        """-> (draw buffer, its layout, deferred-check words) of the capture that has just ended; this module forgets them, so that an eager call
        allocates its own and never refills or reallocates what a graph's copy nodes point at."""
        got = (getattr(self, "_pinned_draws", None), getattr(self, "_pinned_draws_layout", None), getattr(self, "_deferred_host", None))
        self._pinned_draws = None
        self._pinned_draws_event = None
        self._deferred_host = None
        self._deferred = None
        return got

    def _train_fused_ok(self, scene, net, lean):
        """The fused TRAINING path of render_core: K17 (net) + K18 on a device-side selection (ops.StepPoints), no host synchronisation."""
        if lean or net is None or not self.fused_train or not torch.is_grad_enabled():
            return False
        nf = len(scene.views.feat_tex)
        return (ops.BlendPlan.supported(self.color_network) and nf <= 5 and self.color_network.ray_dir_fc[2].weight.shape[0] == 3 + 4 * nf
                and scene.views.nv >= 2)

    def _render_core_train(self, rays_o, rays_d, z_vals, sample_dist, scene, intrs, c2ws, cos_anneal_ratio, step, net, pts_random, extra_pts,
                           extra_valid):
        """render_core (:152-349) in training mode on the fused kernels, with the masked evaluation's selection left on the device:
        [ray samples | random points | extra (pseudo) points] share ONE K17 forward / backward pair; dense outputs carry the reference's
        defaults for unselected rows (Q8).  extra_valid: uint8 flags of extra_pts, or None (all of them are evaluated)."""
        b, n = z_vals.shape
        dev = z_vals.device
        n_ray, n_r = b * n, pts_random.shape[0]
        n_x = 0 if extra_pts is None else extra_pts.shape[0]
        total = n_ray + n_r + n_x
        pts_all = torch.empty(total, 3, device=dev, dtype=torch.float32)
        valid_all = torch.empty(total, device=dev, dtype=torch.uint8)
        ops.ray_points(rays_o, rays_d, z_vals, scene.masks, mid=True, sample_dist=sample_dist, out=(pts_all[:n_ray], valid_all[:n_ray]))
        pts_all[n_ray:n_ray + n_r].copy_(pts_random)
        if n_x:
            pts_all[n_ray + n_r:].copy_(extra_pts)
            if extra_valid is None:
                valid_all[n_ray + n_r:].fill_(1)
            else:
                valid_all[n_ray + n_r:].copy_(extra_valid)
        s_views = scene.views.nv - 1
        sel = ops.StepPoints(pts_all, valid_all, n_ray, n_r, s_views, z=z_vals, variance=self.deviation_network.variance)
        with_tv = ops.tv_levels_ok(scene.volumes, scene.mask_volumes)       # the regulariser rides on the network's Function: one gradient buffer
        y_all, g_all, s_all, *tv_reg = net(pts_all, sel, tv=with_tv)
        sampled_color, src_vis = ops.blend_train(self.color_network, scene.views, pts_all, sel)
        comp = ops.composite_train(sel, rays_o, rays_d, z_vals, sample_dist, y_all, g_all, s_all, sampled_color, self.deviation_network.variance,
                                   valid_all[:n_ray], sel.vis, cos_anneal_ratio, scene.ref_rotation())
        out = {
            "color_fine": comp["color"],
            "render_depth": comp["depth"],
            "normal": comp["normal"],
            "weights": comp["weights"],
            "weight_sum": comp["wsum"][:, None],
            "weight_max": comp["wmax"][:, None],
            "inside_sphere": comp["inside"],
            "valid_mask": comp["valid"].bool()[:, None],
            "mid_inside_sphere": comp["mid_in"][:, None],
            "sdf_depth": comp["sdf_depth"][:, None],
            "gradients": g_all[:n_ray].reshape(b, n, 3),
            "s_val": sel.scalars[2:3].detach().reshape(1, 1).expand(b * n, 1),
            "gradient_error": comp["gradient_error"],
            "smooth_error": comp["smooth_error"],
            "sparse_sdf": torch.cat([y_all[n_ray:n_ray + n_r], y_all[:n_ray]]),
            "tv_reg": tv_reg[0] if with_tv else self.tv_regularization(scene.volumes, scene.mask_volumes),
        }
        if n_x:
            out["_extra_sdf_dense"] = y_all[n_ray + n_r:]
        self._last_step_counts = sel.counts            # (device; forward() hands it to the deferred checks)
        # surface point of the first sign change (written by the compositing launch), its SDF gradient (the reference builds the second-order
        # graph here too and throws it away: the normal is used detached, :306-310), and the plane-induced patch warp in ONE launch
        g0 = net.first_order(comp["pts_cross"])
        warp = scene.warp_features(use_match=not (step is None or step < 5))
        out["ref_gray_val"], out["sampled_gray_val"] = ops.patch_warp(comp["z_cross"], rays_o, rays_d, g0, scene.views.cams, warp)
        return out

    def render_core(self, rays_o, rays_d, z_vals, sample_dist, volumes, mask_volumes, features, match_features, imgs, intrs, c2ws,
                    cos_anneal_ratio, step, scene=None, lean=False, pts_random=None, extra_pts=None, net=None, extra_valid=None):
        """Everything after sampling (:152-349).  `lean` (validate only) skips the quantities validate discards:
        second derivatives, random-point SDF, TV, the surface-point gradient and the patch warp.
        extra_pts: more points whose SDF the caller wants from the same network pass (forward()'s pseudo points, :484-497):
        returned under the private key "_extra_sdf" -- or, with extra_valid (their uint8 mask flags, left on the device), as the dense
        "_extra_sdf_dense" (zeros where the flag is clear, the reference's pseudo_sdf) of the fused training path."""
        if scene is None:
            scene = Scene(volumes, mask_volumes, imgs, features, match_features, intrs, c2ws)
        b, n = z_vals.shape
        dev = z_vals.device
        if net is None:                                # training: one fused evaluator for every SDF query of this step
            net = self._train_net(scene, lean)
        if self._train_fused_ok(scene, net, lean) and (features is not None):
            if pts_random is None:
                pts_random = self._host_draws(dev)[1]                                      # CPU generator, :256
            return self._render_core_train(rays_o, rays_d, z_vals, sample_dist, scene, intrs, c2ws, cos_anneal_ratio, step, net, pts_random,
                                           extra_pts, extra_valid)
        if extra_valid is not None:                    # the generic path evaluates the selected extra points only
            extra_pts = extra_pts[torch.nonzero(extra_valid)[:, 0]]
        need_vol_grad = torch.is_grad_enabled() and any(v.requires_grad for v in scene.volumes)
        vols = scene.volumes if need_vol_grad else scene.volumes_nograd()

        cached = getattr(self, "_mid_points", None)
        self._mid_points = None
        if cached is not None and cached[0] is z_vals and cached[1] == float(sample_dist):      # _sample_rays' last launch already made them
            pts, valid = cached[2], cached[3]
        else:
            pts, valid = ops.ray_points(rays_o, rays_d, z_vals, scene.masks, mid=True, sample_dist=sample_dist)
        plan = self._fused_plan(vols) if lean else None
        bplan = self._fused_blend_plan(scene.views) if lean else self._fused_blend_plan(scene.views, features, imgs)
        sdf_random = extra_sdf = None
        if plan is not None and bplan is not None:     # fully fused inference: nothing in this branch synchronises with the host
            sdf, gradients = torch.empty(b * n, 1, device=dev), torch.empty(b * n, 3, device=dev)
            sampled_color = torch.empty(b * n, 3, device=dev)
            src_vis = torch.empty(b * n, scene.views.nv - 1, device=dev, dtype=torch.uint8)
            idx, count = ops.compact_fill(valid, sdf=sdf, grad=gradients, rgb=sampled_color, vis=src_vis)      # (defaults of the unselected rows, Q8)
            ops.sdf_mlp(plan, vols, pts, index=idx, want_grad=True, sdf_out=sdf, grad_out=gradients, precision=self._precision(plan, True), count=count)
            smooth = None
            ops.blend_views(bplan, scene.views, pts, index=idx, rgb_out=sampled_color, vis_out=src_vis, count=count)
        else:
            idx = self._select(valid)
            pts_v = pts[idx]
            if plan is not None:                       # fused look-up + MLP + d/dx, scattered straight into the dense arrays
                sdf = torch.full((b * n, 1), 100.0, device=dev)
                gradients = torch.zeros(b * n, 3, device=dev)
                ops.sdf_mlp(plan, vols, pts, index=idx, want_grad=True, sdf_out=sdf, grad_out=gradients, precision=self._precision(plan, True))
                smooth = None
            else:
                if lean:
                    with torch.enable_grad():
                        x = pts_v.clone().requires_grad_(True)
                        sdf_v = self.sdf_network.sdf(x, vols)
                        grad_v = torch.autograd.grad(sdf_v, x, torch.ones_like(sdf_v))[0]
                    sdf_v, smooth_v = sdf_v.detach(), None
                elif net is not None:                      # K17: the ray samples, the 1024 random points (:256-257) and the caller's extra
                    if pts_random is None:                 # points share ONE forward / backward pair of launches
                        pts_random = torch.rand([N_RANDOM_PTS, 3]).to(dev) * 2 - 1             # CPU generator, :256
                    batch = [pts_v, pts_random] + ([extra_pts] if extra_pts is not None else [])
                    y_all, g_all, s_all = net(torch.cat(batch))
                    n_v, n_r = pts_v.shape[0], pts_random.shape[0]
                    sdf_v, grad_v, smooth_v = y_all[:n_v], g_all[:n_v], s_all[:n_v]
                    sdf_random = y_all[n_v:n_v + n_r]
                    extra_sdf = y_all[n_v + n_r:] if extra_pts is not None else None
                else:                                      # one forward pass for :179 (sdf) and :188 (gradient, smooth)
                    sdf_v, grad_v, smooth_v = self.sdf_network.sdf_gradient_smooth(pts_v.clone(), vols)
                sdf = torch.full((b * n, 1), 100.0, device=dev).index_put((idx,), sdf_v)
                gradients = torch.zeros(b * n, 3, device=dev).index_put((idx,), grad_v)
                smooth = None if smooth_v is None else torch.zeros(b * n, 3, device=dev).index_put((idx,), smooth_v)
            if bplan is not None:                      # K4 + colour network fused, scattered into the dense arrays
                sampled_color, src_vis = ops.blend_views(bplan, scene.views, pts, index=idx)
            elif (self.fused_train and torch.is_grad_enabled() and ops.BlendPlan.supported(self.color_network) and len(scene.views.feat_tex) <= 5
                  and self.color_network.ray_dir_fc[2].weight.shape[0] == 3 + 4 * len(scene.views.feat_tex)):
                color_v, vis_v = ops.blend_train(self.color_network, scene.views, pts_v)      # K18: forward / backward in one launch each
                sampled_color = torch.zeros(b * n, 3, device=dev).index_put((idx,), color_v)
                src_vis = torch.zeros(b * n, vis_v.shape[1], dtype=torch.bool, device=dev).index_put((idx,), vis_v)
            else:
                feat_views, ray_diff, vis_v = lookup_feature(pts_v, imgs, intrs, c2ws, features, views=scene.views)
                color_v = self.color_network(feat_views, ray_diff, vis_v)
                sampled_color = torch.zeros(b * n, 3, device=dev).index_put((idx,), color_v)
                src_vis = torch.zeros(b * n, vis_v.shape[1], dtype=torch.bool, device=dev).index_put((idx,), vis_v)

        inv_s = self.deviation_network(torch.zeros([1, 3], device=dev))[:, :1].clip(1e-6, 1e6)
        comp = ops.composite(rays_o, rays_d, z_vals, sample_dist, sdf, gradients, smooth, sampled_color, valid, src_vis, inv_s,
                             cos_anneal_ratio, scene.ref_rotation())
        gradients = gradients.reshape(b, n, 3)
        out = {
            "color_fine": comp["color"],
            "render_depth": comp["depth"],
            "normal": comp["normal"],
            "weights": comp["weights"],
            "weight_sum": comp["wsum"][:, None],
            "weight_max": comp["wmax"][:, None],
            "inside_sphere": comp["inside"],
            "valid_mask": comp["valid"].bool()[:, None],
            "mid_inside_sphere": comp["mid_in"][:, None],
            "sdf_depth": comp["sdf_depth"][:, None],
            "gradients": gradients,
            "s_val": (1.0 / inv_s).expand(b * n, 1),
            "gradient_error": comp["eik_num"].sum() / (comp["eik_den"].sum() + 1e-5),
        }
        if lean:
            return out
        out["smooth_error"] = torch.linalg.norm(comp["smooth_vec"], ord=2, dim=-1).abs().mean()

        if pts_random is None:
            pts_random = torch.rand([N_RANDOM_PTS, 3]).to(dev) * 2 - 1                     # CPU generator, :256
        if sdf_random is None:
            sdf_random = self.sdf_network.sdf(pts_random, vols)
        out["sparse_sdf"] = torch.cat([sdf_random, sdf])
        if extra_pts is not None:
            out["_extra_sdf"] = extra_sdf if extra_sdf is not None else self.sdf_network.sdf(extra_pts, vols)
        out["tv_reg"] = self.tv_regularization(scene.volumes, scene.mask_volumes)

        # surface point of the first sign change and the plane-induced patch warp (:288-328)
        pts_sdf0 = rays_o[:, None, :] + rays_d[:, None, :] * comp["z_cross"][:, None, None]
        # the reference builds the second-order graph here too and throws it away: the normal is used detached (:306-310)
        if net is not None:
            g0 = net.first_order(pts_sdf0)
        else:
            g0, _ = self.sdf_network.gradient(pts_sdf0.detach().reshape(-1, 3).clone(), vols, second_order=False)
        g0 = g0.reshape(b, 1, 3)
        g0_norm = torch.linalg.norm(g0, ord=2, dim=-1, keepdim=True)
        g0 = g0 / torch.where(g0_norm <= 0, torch.full_like(g0_norm, 1e-8), g0_norm)
        normals_ref = (g0 @ c2ws[0, :3, :3]).detach()                                       # R^T n as row vectors
        warp = scene.warp_features(use_match=not (step is None or step < 5))
        out["ref_gray_val"], out["sampled_gray_val"] = surface_patch_warp(pts_sdf0, normals_ref, warp, intrs, c2ws)
        return out

    # ----------------------------------------------------------------------------------------------------------
    # render
    # ----------------------------------------------------------------------------------------------------------
    def _coarse_steps(self, dev):
        """torch.linspace(0, 1, n_samples) as the reference computes it (on the host, implicit_surface.py:357), uploaded ONCE per
        device: a pageable host-to-device copy waits for the stream to drain on ROCm, which put the host in lock-step with the GPU
        once per ray chunk (10 ms of idle GPU per 480x640 image)."""
        cached = getattr(self, "_steps_cache", None)
        if cached is None or cached[0] != self.n_samples or cached[1].device != dev:
            cached = (self.n_samples, torch.linspace(0.0, 1.0, self.n_samples).to(dev))
            self._steps_cache = cached
        return cached[1]

    def render(self, rays_o, rays_d, near, far, volumes, mask_volumes, imgs, features, match_features, intrs, c2ws, cos_anneal_ratio, step,
               scene=None, lean=False, t_rand=None, pts_random=None, extra_pts=None, extra_valid=None):
        if scene is None:
            scene = Scene(volumes, mask_volumes, imgs, features, match_features, intrs, c2ws)
        b = len(rays_o)
        dev = rays_o.device
        net = self._train_net(scene, lean)
        if self.perturb > 0 and t_rand is None and pts_random is None and self._train_fused_ok(scene, net, lean):
            t_rand, pts_random = self._host_draws(dev, b)          # both host draws of the step (:362, then :256) through one pinned copy
        rays_o, rays_d = rays_o.float().contiguous(), rays_d.float().contiguous()
        sample_dist = 2.0 / self.n_samples                                                  # unit-sphere assumption (:355)
        steps = self._coarse_steps(dev)
        if self.perturb > 0 and t_rand is None:
            t_rand = torch.rand([b, 1])                                                     # CPU generator, :362
        z_vals = ops.coarse_z(near, far, steps, t_rand.to(dev, non_blocking=True) if self.perturb > 0 else None, b)      # (:356-363, one launch)
        if self.n_importance > 0:
            # (the fused TRAINING path writes its mid-points into slices of the step's point array itself)
            z_vals = self._sample_rays(rays_o, rays_d, z_vals, scene, net, mid_points=None if self._train_fused_ok(scene, net, lean) else sample_dist)
        return self.render_core(rays_o, rays_d, z_vals, sample_dist, volumes, mask_volumes, features, match_features, imgs, intrs, c2ws,
                                cos_anneal_ratio, step, scene=scene, lean=lean, pts_random=pts_random, extra_pts=extra_pts, net=net,
                                extra_valid=extra_valid)

    # ----------------------------------------------------------------------------------------------------------
    # geometry + validation
    # ----------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def sdf_grid(self, volumes, bound_min, bound_max, resolution, chunk=1 << 21, shard=None):
        """u = -sdf on the resolution^3 lattice (:407-421), kept on the device.  shard (gens_amd.distributed.Shard): this rank evaluates
        the chunks `index mod world` and the slabs are gathered on every rank (None under Shard.single until the last shard arrives)."""
        vols = volumes if isinstance(volumes, ops.VolumeSet) else ops.VolumeSet.packed(volumes)
        dev = vols.tensors[0].device
        total = resolution ** 3
        n_chunks = -(-total // chunk)
        own = range(n_chunks) if shard is None else shard.chunks(n_chunks)
        u = torch.zeros(len(own), chunk, device=dev) if shard is not None else torch.empty(total, device=dev)
        split_half = None                              # a value outside the half range: the lattice again in float32, like the image
        for attempt in range(2):
            for k, c in enumerate(own):
                first = c * chunk
                count = min(chunk, total - first)
                pts = ops.lattice_points(bound_min.tolist(), bound_max.tolist(), resolution, first, count, dev)
                plan = self._fused_plan(vols)
                prec = "f32" if split_half is False else self._precision(plan) if plan is not None else "f32"
                sdf = ops.sdf_mlp(plan, vols, pts, precision=prec) if plan is not None else self.sdf_network.sdf(pts, vols)
                if shard is not None:
                    u[k, :count] = -sdf[:, 0]
                else:
                    u[first:first + count] = -sdf[:, 0]
            overflowed = split_half is not False and self._split_half_overflowed()
            if shard is not None and split_half is not False and self.sdf_precision == "f16x2":
                overflowed = shard.any(overflowed)
            if not overflowed:
                break
            split_half = False
        if shard is not None:
            u = shard.gather_chunks(u, n_chunks)
            if u is None:
                return None
            u = u.reshape(-1)[:total]
        return u.reshape(resolution, resolution, resolution)

    def extract_geometry(self, volumes, bound_min, bound_max, resolution, threshold, shard=None):
        """-> vertices (V,3) float64, triangles (T,3) int32 as numpy arrays (implicit_surface.py:407-427).  The SDF lattice and
        the marching cubes both run on the device (the reference: 512 D2H copies + PyMCubes on the host); only the mesh is copied."""
        u = self.sdf_grid(volumes, bound_min, bound_max, resolution, shard=shard)
        if u is None:
            return None, None
        vertices, triangles = ops.marching_cubes(u, threshold)
        vertices, triangles = vertices.cpu().numpy(), triangles.cpu().numpy()
        b_max, b_min = bound_max.detach().cpu().numpy(), bound_min.detach().cpu().numpy()
        vertices = vertices / (resolution - 1.0) * (b_max - b_min)[None, :] + b_min[None, :]
        return vertices, triangles

    @torch.no_grad()
    def validate(self, rays_o, rays_d, near, far, volumes, mask_volumes, imgs, features, match_features, intrs, c2ws, bound_min, bound_max,
                 hw, cos_anneal_ratio=1.0, step=None, extract_geometry=True, mesh_resolution=512, threshold=0.0, scene=None, shard=None):
        """shard (gens_amd.distributed.Shard, optional): render only this rank's contiguous ray range and evaluate only its lattice
        chunks; the (P, 8) image buffer / the lattice slabs are gathered over RCCL, so every rank returns the whole image.  The jitter
        of EVERY ray is drawn on every rank from the identically seeded CPU generator (the reference's draw order) and sliced with
        the rays: the image does not depend on the partition."""
        outputs = {}
        if scene is None:
            scene = Scene(volumes, mask_volumes, imgs, features, match_features, intrs, c2ws)
        height, width = int(hw[0]), int(hw[1])
        n_rays = rays_o.shape[0]
        r0, r1 = (0, n_rays) if shard is None else shard.rays(n_rays)
        # the jitter thread starts first: its ~1.5 ms per 32 768 rays (the reference's draw order costs 13 draws per ray) then hide behind
        # the mesh extraction, which draws nothing from the generator
        chunk = self.val_chunk_for(r1 - r0)
        self.last_val_chunk = chunk
        jitter = None
        if self.perturb > 0:
            jitter = self._take_speculated_jitter(n_rays)
            if jitter is None:
                jitter = JitterStream(n_rays, chunk, self._pinned(n_rays, 1, "_pinned_jitter"))
        if extract_geometry:
            import time
            t_geo = time.perf_counter()
            outputs["vertices"], outputs["triangles"] = self.extract_geometry(scene.volumes_nograd(), bound_min, bound_max, mesh_resolution,
                                                                              threshold, shard=shard)
            self.last_geometry_s = time.perf_counter() - t_geo     # (ends with the mesh's read-back: wall time is the item's share; bench.py's default_path)
        # one (P, 8) device buffer [rgb | normal | sdf_depth | render_depth] filled chunk by chunk: ONE D2H copy per image
        # into a pinned host buffer (the reference copies 4 tensors per 256-ray chunk, implicit_surface.py:446-453)
        import time as _time
        t_render = _time.perf_counter()                 # (the image's share of the call, as last_geometry_s is the mesh's: bench.py's default_path)
        image = torch.empty(r1 - r0, 8, device=rays_o.device, dtype=torch.float32)

        def render_image():
            for s in range(r0, r1, chunk):
                e = min(s + chunk, r1)
                r = self.render(rays_o[s:e], rays_d[s:e], near, far, volumes, mask_volumes, imgs, features, match_features, intrs, c2ws,
                                cos_anneal_ratio, step, scene=scene, lean=True, t_rand=None if jitter is None else jitter.slice(s, e))
                image[s - r0:e - r0, 0:3] = r["color_fine"]
                image[s - r0:e - r0, 3:6] = (r["gradients"] * r["weights"][..., None] * r["inside_sphere"][..., None]).sum(dim=1)
                image[s - r0:e - r0, 6] = r["sdf_depth"].reshape(-1)
                image[s - r0:e - r0, 7] = r["render_depth"].reshape(-1)

        render_image()
        if jitter is not None:
            jitter.join()
            if jitter.generator is not None:           # the head start was used: its draws are the default generator's past now
                torch.set_rng_state(jitter.end_state())
        overflowed = self._split_half_overflowed()
        if shard is not None and self.sdf_precision == "f16x2":
            overflowed = shard.any(overflowed)         # every rank re-renders or none does: the gathered image never mixes precisions
        if overflowed:                                 # a value left the half range: the same image, same jitter, in float32
            saved, self.sdf_precision = self.sdf_precision, "f32"
            try:
                render_image()
            finally:
                self.sdf_precision = saved
        if shard is not None:               # RCCL all_gather of the rendered buffers (config 4), device to device
            image = shard.gather_rows(image, n_rays, key="image")
            if image is None:               # Shard.single: the other shards have not been rendered yet
                return None
        self.last_device_image = image      # (P, 8) [rgb | normal | sdf_depth | render_depth] on the device: what a multi-GPU driver gathers
        # the 8-bit-range images (implicit_surface.py:455-463: colour * 256, rot @ normal * 128 + 128, clipped) are formed on the DEVICE and
        # travel in the same single copy: the host only reshapes (on 307 200 pixels the numpy versions cost milliseconds with the GPU idle)
        # and laid out as five CONTIGUOUS blocks [rgb (P,3) | img_fine (P,3) | normal_img (P,3) | sdf_depth (P) | render_depth (P)], so the host
        # makes one plain copy out of the reused page-locked buffer and hands out views of it (strided column copies cost 4.6 ms of idle GPU per image)
        p_ = image.shape[0]
        rot = scene.views.cams.rot_inv.view(3, 3)
        post = torch.empty(11 * p_, device=image.device, dtype=image.dtype)
        post[0:3 * p_].view(p_, 3).copy_(image[:, 0:3])
        torch.clamp(image[:, 0:3] * 256, 0, 255, out=post[3 * p_:6 * p_].view(p_, 3))
        n0, n1, n2 = image[:, 3:4], image[:, 4:5], image[:, 5:6]
        torch.clamp(((n0 * rot[:, 0] + n1 * rot[:, 1]) + n2 * rot[:, 2]) * 128 + 128, 0, 255, out=post[6 * p_:9 * p_].view(p_, 3))   # rot @ n per pixel
        post[9 * p_:10 * p_].copy_(image[:, 6])
        post[10 * p_:11 * p_].copy_(image[:, 7])
        host = self._pinned(11 * n_rays, 1, "_pinned_post")
        host.copy_(post.view(-1, 1), non_blocking=True)
        if self.perturb > 0 and self.speculate_jitter:
            self._speculate_jitter(n_rays, chunk)      # image after image (runner.py:215's loop): the next image's draws start while this one drains
        status = scene.views.cams.status.cpu()         # (rides behind the image copy: a singular camera matrix raises, as torch.inverse does)
        torch.cuda.current_stream().synchronize()
        if int(status) != 0:
            raise RuntimeError("linalg.inv: a camera pose or intrinsics matrix of the scene is singular")
        flat = host.numpy().reshape(-1).copy()
        outputs["color_fine"] = torch.from_numpy(flat[0:3 * p_].reshape(p_, 3))
        outputs["img_fine"] = flat[3 * p_:6 * p_].reshape([height, width, 3])
        outputs["normal_img"] = flat[6 * p_:9 * p_].reshape([height, width, 3])
        outputs["sdf_depth"] = flat[9 * p_:10 * p_].reshape([height, width])
        outputs["render_depth"] = flat[10 * p_:11 * p_].reshape([height, width])
        self.last_render_s = _time.perf_counter() - t_render
        return outputs

    def val_chunk_for(self, n_rays):
        """Rays per render() chunk for a range of n_rays rays: `val_chunk` when the caller set one, else equal chunks of at most 32 768."""
        from ...chunking import balanced_chunk
        return int(self.val_chunk) if self.val_chunk else balanced_chunk(n_rays)

    def _speculate_jitter(self, n_rays, chunk=None):
        """Draw the NEXT validate() call's jitter now, on a helper thread with a PRIVATE generator that starts at the default CPU generator's
        current state.  The reference's draw order costs 13 generator draws per ray -- 14 ms of host time for a 480 x 640 image -- and a rank
        that renders the LAST rays of a ray-sharded image needs the state behind all the others': without the head start the first chunk waits
        1.5 ms per image for its draws, the last rank of eight ~12 ms of a 35 ms step.  The next validate() uses the draws only if it asks for
        the same number of rays AND the default generator still stands where the draws started (nobody drew in between: image after image, as
        runner.py:215's loop renders them); it then moves the default generator behind them.  Otherwise they are dropped -- a training step
        between two validations sees exactly the reference's stream, whatever was drawn ahead."""
        pending = getattr(self, "_jitter_ahead", None)
        state = torch.get_rng_state()
        if pending is not None:
            if pending[0] == n_rays and torch.equal(pending[1], state):
                return                                 # already drawing exactly this
            pending[2].join()                          # (its page-locked buffer is about to be reused)
        g = torch.Generator()
        g.set_state(state)
        # two page-locked buffers take turns: the image being rendered reads one while the next image's draws fill the other
        self._jitter_parity = 1 - getattr(self, "_jitter_parity", 0)
        buf = self._pinned(n_rays, 1, "_pinned_jitter_ahead%d" % self._jitter_parity)
        self._jitter_ahead = (n_rays, state, JitterStream(n_rays, chunk or self.val_chunk_for(n_rays), buf, generator=g))

    def prefetch_jitter(self, n_rays):
        """Rounds 2 - 5's explicit head start for drivers that render image after image; validate() now does it itself (_speculate_jitter).
        Kept for callers that want the draws started before the FIRST image."""
        if self.perturb > 0:
            self._speculate_jitter(n_rays)

    def _take_speculated_jitter(self, n_rays):
        pending = getattr(self, "_jitter_ahead", None)
        self._jitter_ahead = None
        if pending is None:
            return None
        if pending[0] != n_rays or not torch.equal(pending[1], torch.get_rng_state()):
            pending[2].join()                          # drawn for nothing -- on its own generator: the default one never moved
            return None
        return pending[2]

    def join_speculation(self):
        """Wait for a head start that will not be used (before interpreter shutdown in scripts that time themselves)."""
        pending = getattr(self, "_jitter_ahead", None)
        if pending is not None:
            pending[2].join()

    def _pinned(self, n_rays, cols=8, slot="_pinned_image"):
        """Page-locked (P, cols) staging buffer (rendered image / ray jitter), kept between validate() calls, which end with a
        stream synchronisation, so reuse is safe."""
        buf = getattr(self, slot, None)
        if buf is None or buf.shape[0] != n_rays:
            buf = torch.empty(n_rays, cols, dtype=torch.float32, pin_memory=True)
            setattr(self, slot, buf)
        return buf

    def _defer_checks(self, counts, cam_status):
        """Host-side checks of a fused training step, taken off its critical path: the counts of gens_compact_points and the scene set-up's
        status word travel to page-locked memory behind the step's launches."""
        host = getattr(self, "_deferred_host", None)
        if host is None:
            host = self._deferred_host = torch.zeros(4, dtype=torch.int32, pin_memory=True)
        host[:3].copy_(counts, non_blocking=True)
        host[3:4].copy_(cam_status, non_blocking=True)
        if torch.cuda.is_current_stream_capturing():         # (a captured step: the copies are graph nodes; the replaying loop checks after its own synchronisation)
            self._deferred = None
            return
        ev = torch.cuda.Event()
        ev.record()
        self._deferred = ev

    def check_deferred(self):
        """Raise what the last fused training step would have raised in the reference (no valid pseudo point, implicit_surface.py:494-495; a
        singular camera matrix, torch.inverse).  Called at the start of every forward().  The reference raises INSIDE the bad step's forward
        (before its backward and optimiser step); here the bad step's pseudo-point term is zero (value and gradient) and the error arrives one
        call later, so a loop that must not apply the bad step's update -- or that ends -- calls this itself once the step's forward has run:
        after the loss read-back (free: the read-back synchronised) and before optimizer.step(), as distributed.FinetuneStepper does."""
        ev = getattr(self, "_deferred", None)
        if ev is None or torch.cuda.is_current_stream_capturing():
            return
        self._deferred = None
        ev.synchronize()
        self.check_deferred_host()

    def check_deferred_host(self):
        """The same checks on what the last step left in page-locked memory; the caller has synchronised with that step (a graph replay
        followed by the loss read-back)."""
        host = getattr(self, "_deferred_host", None)
        if host is None:
            return
        if int(host[3]) != 0:
            raise RuntimeError("linalg.inv: a camera pose or intrinsics matrix of the previous step's scene is singular")
        if int(host[2]) < 1:
            raise RuntimeError("No valid pseudo pts!")

    # inputs of a training step that change from call to call (copied into a captured step's static tensors); everything else in `ipts` is
    # either consumed by the caller's loss (colour, depths, masks) or describes the item (scene, file names)
    STEP_INPUTS = ("imgs", "intrs", "c2ws", "rays_o", "rays_d", "near", "far", "pseudo_pts")

    def _auto_graph_ok(self, mode, ipts, volumes, features):
        """May this call run as a captured step (gens_amd.graph.AutoGraph)?  Training / fine-tune steps on the device that take the fused path
        (K17 + K18 on a device-side selection: nothing in them is read back by the host)."""
        from ... import graph
        if mode == "val" or not getattr(self, "auto_graph", True) or not graph.auto_graph_enabled() or not torch.is_grad_enabled():
            return False
        if getattr(self, "_auto_suppressed", False):          # an enclosing GenS.forward manages this step
            return False
        rays = ipts["rays_o"]
        if not (torch.is_tensor(rays) and rays.is_cuda) or torch.cuda.is_current_stream_capturing():
            return False
        if not self.fused_train or self.n_importance <= 0 or features is None or ipts["imgs"].shape[0] < 2:
            return False
        if not any(p.requires_grad for p in self.sdf_network.parameters()) and not any(v.requires_grad for v in volumes):
            return False
        nf = len(features)
        if not (ops.BlendPlan.supported(self.color_network) and nf <= 5 and self.color_network.ray_dir_fc[2].weight.shape[0] == 3 + 4 * nf):
            return False
        return ops.SdfTrainStep.supported(self.sdf_network, len(volumes)) and all(v.dim() == 5 and v.shape[1] == 4 for v in volumes)

    def forward(self, mode, ipts, volumes, mask_volumes, features, match_features, cos_anneal_ratio=1.0, step=None):
        """implicit_surface.py:472-499.  A training / fine-tune call that qualifies (_auto_graph_ok) runs as a CAPTURED step after two eager
        ones: forward and backward are one HIP graph replay each (gens_amd.graph.AutoGraph), behind this unchanged signature -- the loop of
        runner.py:157-166 stays as it is.  `self.auto_graph = False` or GENS_AUTO_GRAPH=0: every call eager."""
        if not self._auto_graph_ok(mode, ipts, volumes, features):
            return self._forward_impl(mode, ipts, volumes, mask_volumes, features, match_features, cos_anneal_ratio, step)
        from ... import graph
        auto = getattr(self, "_auto", None)
        if auto is None:
            auto = self._auto = graph.AutoGraph()
        copied, refs, layout = {}, [], []

        def take(name, t):
            """persistent tensors (parameters, leaves the caller optimises) are used where they are; the rest is copied per call"""
            if isinstance(t, nn.Parameter) or (t.is_leaf and t.requires_grad):
                refs.append(t)
                layout.append((name, len(refs) - 1))
            else:
                copied[name] = t
                layout.append((name, None))
        for k in self.STEP_INPUTS:
            if k in ipts:
                take(k, ipts[k])
        same_match = all(a is b for a, b in zip(features, match_features)) and len(features) == len(match_features)
        groups = [("volumes", volumes), ("mask_volumes", mask_volumes), ("features", features)] + ([] if same_match else [("match_features", match_features)])
        for gname, group in groups:
            for i, t in enumerate(group):
                take(f"{gname}.{i}", t)
        params = [p for p in self.parameters()]
        n_in = len(refs)
        refs = refs + params
        use_match = not (step is None or step < 5)
        counts = tuple(len(g) for _, g in groups)

        where = dict(layout)

        def body(cp, sc, alias):
            def get(name):
                return cp[name] if name in cp else alias(refs[where[name]])
            ip = dict(ipts)
            for k in self.STEP_INPUTS:
                if k in ipts:
                    ip[k] = get(k)
            lists = {gname: [get(f"{gname}.{i}") for i in range(len(group))] for gname, group in groups}
            feats = lists["features"]
            return self._forward_impl(mode, ip, lists["volumes"], lists["mask_volumes"], feats, feats if same_match else lists["match_features"],
                                      sc["cos_anneal_ratio"], 5.0 if use_match else 0.0)
        key = ("ImplicitSurface", mode, use_match, same_match, counts, tuple(n for n, _ in layout), n_in, self.training)
        return auto.run(key, copied, refs, {"cos_anneal_ratio": float(cos_anneal_ratio)}, body, [self], module=self)

    def _forward_impl(self, mode, ipts, volumes, mask_volumes, features, match_features, cos_anneal_ratio=1.0, step=None):
        imgs, intrs, c2ws = ipts["imgs"], ipts["intrs"], ipts["c2ws"]
        rays_o, rays_d, near, far = ipts["rays_o"], ipts["rays_d"], ipts["near"], ipts["far"]
        scene = Scene(volumes, mask_volumes, imgs, features, match_features, intrs, c2ws)
        self.check_deferred()                          # what the previous fused training step left to verify (see below)
        pseudo_pts = idx = None
        fused = mode != "val" and "pseudo_pts" in ipts and self._train_fused_ok(scene, self._train_net(scene, False), False)
        if fused:
            # (:484-497) on the fused training path the pseudo points ride on the render's network pass with their mask flags LEFT ON THE
            # DEVICE: nothing is read back in the middle of the step.  The reference's `raise "No valid pseudo pts!"` (:494-495) needs the
            # count on the host; it travels behind the step through a page-locked copy and is raised by check_deferred() -- at the next
            # forward() at the latest (such a step's pseudo_sdf is all zeros, so its loss term and gradient are zero).
            pseudo_pts = ipts["pseudo_pts"].float().contiguous()
            flags = torch.empty(pseudo_pts.shape[0], device=pseudo_pts.device, dtype=torch.uint8)
            ops.lookup_mask(pseudo_pts, scene.masks, out=flags)
            outputs = self.render(rays_o, rays_d, near, far, volumes, mask_volumes, imgs, features, match_features, intrs, c2ws,
                                  cos_anneal_ratio, step, scene=scene, extra_pts=pseudo_pts, extra_valid=flags)
            outputs["pseudo_sdf"] = outputs.pop("_extra_sdf_dense")
            self._defer_checks(self._last_step_counts, scene.views.cams.status)
            return outputs
        host = getattr(self, "_deferred_host", None)
        if host is not None and not torch.cuda.is_current_stream_capturing():
            # this step leaves nothing to verify later (it raises on the spot, or carries no pseudo points): what an EARLIER fused step parked in
            # page-locked memory was checked two lines up and must not be read again by check_deferred_host() / GraphedStep.check()
            host[2], host[3] = 1, 0
        if "pseudo_pts" in ipts:                       # (:484-497) the mask look-up draws nothing from the generator, so it can come first
            pseudo_pts = ipts["pseudo_pts"].float()
            valid = ops.lookup_mask(pseudo_pts, scene.masks)
            if int(valid.sum()) < 1:
                raise RuntimeError("No valid pseudo pts!")                                  # the reference raises a str (:494-495)
            idx = torch.nonzero(valid)[:, 0]
        if mode == "val":
            outputs = self.validate(rays_o, rays_d, near, far, volumes, mask_volumes, imgs, features, match_features, intrs, c2ws,
                                    ipts["bound_min"], ipts["bound_max"], ipts["hw"], cos_anneal_ratio, step, scene=scene)
            extra = None
        else:                                          # the pseudo points ride on the render's network pass
            outputs = self.render(rays_o, rays_d, near, far, volumes, mask_volumes, imgs, features, match_features, intrs, c2ws,
                                  cos_anneal_ratio, step, scene=scene, extra_pts=None if idx is None else pseudo_pts[idx])
            extra = outputs.pop("_extra_sdf", None)
        if pseudo_pts is not None:
            if extra is None:
                vols = scene.volumes if any(v.requires_grad for v in scene.volumes) else scene.volumes_nograd()
                extra = self.sdf_network.sdf(pseudo_pts[idx], vols)
            outputs["pseudo_sdf"] = torch.zeros_like(pseudo_pts[:, :1]).index_put((idx,), extra)
        return outputs
