"""K7 (fused source-view look-up + BlendingNetwork in inference) and K18 (the same for a training step).

Part of gens_amd.ops (see ops/__init__.py); citations are relative to /root/reference."""
from .base import *  # noqa: F401,F403
from .sdf import _pack_b16, _pack_b_groups, _poison

# ------------------------------------------------------------------------------------------------------------------
# K7  fused source-view look-up + BlendingNetwork (inference)   (projector.py:278-349 + blending_network.py:69-118)
# ------------------------------------------------------------------------------------------------------------------
def _pad32(b):
    out = torch.zeros(32 * ((b.numel() + 31) // 32), device=b.device, dtype=_f32)
    out[:b.numel()] = b.reshape(-1)
    return out


def _pack_blend_t(layers, n_feat):
    """Weight stream and tables of gens_blend_views_t (k7t_blend.hip).  Activations live in "quad layout" (feature f = 4 kq + q: register
    kq of lane group q); an A fragment of (M tile T, group g of four K quads) holds for lane (m, qk) and j = 0..3 the weight
    W[16 T + 4 (m & 3) + (m >> 2)][slot 16 g + 4 j + qk], so that accumulator register i of lane group q is output feature 4 (4 T + i) + q.
    `layers`: dict name -> (weight, bias).  Returns (stream (G + 2, 64, 4), tab (entries, 4, 8)); the two trailing groups are zero."""
    f = n_feat
    xq = (f + 1) // 4
    dev = layers["rd1"][0].device
    lane = torch.arange(64, device=dev)
    m, qk = lane & 15, lane >> 4
    row_in_tile = 4 * (m & 3) + (m >> 2)

    def product(w, slots, m_tiles, bias=None):
        """w (O, I); slots: list of source columns per input slot (-1: the bias, -2: nothing), padded to whole quads."""
        o = w.shape[0]
        aug = torch.cat([w, (bias if bias is not None else torch.zeros(o, device=dev))[:, None], torch.zeros(o, 1, device=dev)], 1)
        aug = torch.cat([aug, torch.zeros(16 * m_tiles - o, aug.shape[1], device=dev)], 0) if 16 * m_tiles > o else aug
        cols = torch.tensor([c if c >= 0 else (w.shape[1] if c == -1 else w.shape[1] + 1) for c in slots], device=dev)
        nq = (len(slots) + 3) // 4
        cols = torch.cat([cols, torch.full((4 * nq - len(slots),), w.shape[1] + 1, device=dev)])
        groups = []
        for t in range(m_tiles):
            rows = 16 * t + row_in_tile
            for g in range((nq + 3) // 4):
                frag = torch.zeros(64, 4, device=dev, dtype=_f32)
                for j in range(4):
                    kq = 4 * g + j
                    if kq < nq:
                        frag[:, j] = aug[rows, cols[4 * kq + qk]]
                groups.append(frag)
        return groups

    rd1, rd2, b1, b2, v1, v2, u1, u2, r1, r2, r3 = (layers[k] for k in ("rd1", "rd2", "b1", "b2", "v1", "v2", "u1", "u2", "r1", "r2", "r3"))
    xt = (xq + 3) // 4
    g = []
    g += product(rd1[0], [0, 1, 2, 3], 1)
    g += product(rd2[0], list(range(16)), xt)
    pad = [-2] * (4 * xq - f)
    g += product(b1[0], list(range(f)) + pad + list(range(f, 2 * f)) + pad, 4)                         # mean | var, once per point
    g += product(b1[0], list(range(2 * f, 3 * f)) + [-1], 4, b1[1])                                    # x and the bias (slot F)
    g += product(b2[0], list(range(64)), 2)
    g += product(v1[0], list(range(32)), 2)
    g += product(v2[0][:32], list(range(32)), 2)
    g += product(u1[0], list(range(32)), 2)
    g += product(r1[0], list(range(36)) + [36, -1, -2, -2], 1, r1[1])
    g += product(r2[0], list(range(16)), 1)
    stream = torch.stack(g + [torch.zeros(64, 4, device=dev, dtype=_f32)] * 2).contiguous()

    def acc_bias(b, m_tiles):            # [q][4 T + i] = b[16 T + 4 i + q]
        full = torch.zeros(32, device=dev, dtype=_f32)
        full[:min(b.shape[0], 16 * m_tiles)] = b[:16 * m_tiles]
        out = torch.zeros(4, 8, device=dev, dtype=_f32)
        for q in range(4):
            for t in range(m_tiles):
                for i in range(4):
                    out[q, 4 * t + i] = full[16 * t + 4 * i + q]
        return out

    def dot_row(w):                      # [q][kq] = w[4 kq + q]
        full = torch.zeros(32, device=dev, dtype=_f32)
        full[:w.shape[0]] = w
        return full.reshape(8, 4).t().contiguous()

    tab = torch.stack([acc_bias(rd1[1], 1), acc_bias(rd2[1], xt), acc_bias(b2[1], 2), acc_bias(v1[1], 2), acc_bias(v2[1][:32], 2),
                       acc_bias(u1[1], 2), acc_bias(r2[1], 1), dot_row(v2[0][32]), dot_row(u2[0][0]), dot_row(r3[0][0])]).contiguous()
    return stream, tab


class BlendPlan:
    """Weights of a BlendingNetwork re-packed for gens_blend_views (anti_alias_pooling=True, d_feature <= 20)."""

    @staticmethod
    def supported(net):
        return bool(getattr(net, "anti_alias_pooling", False)) and net.base_fc[0].weight.shape[0] == 64 and net.rgb_fc[0].weight.shape[1] == 37

    @staticmethod
    def version(net):
        return tuple(p._version for p in net.parameters()) + tuple(p.data_ptr() for p in net.parameters())

    def __init__(self, net):
        assert BlendPlan.supported(net)
        g = lambda m: (m.weight.detach().to(_f32), m.bias.detach().to(_f32))  # noqa: E731
        with torch.no_grad():
            rd1, rd2 = g(net.ray_dir_fc[0]), g(net.ray_dir_fc[2])
            b1, b2 = g(net.base_fc[0]), g(net.base_fc[2])
            v1, v2 = g(net.vis_fc[0]), g(net.vis_fc[2])
            u1, u2 = g(net.vis_fc2[0]), g(net.vis_fc2[2])
            r1, r2, r3 = g(net.rgb_fc[0]), g(net.rgb_fc[2]), g(net.rgb_fc[4])
            self.n_feat = rd2[0].shape[0]                  # 3 + d_feature
            P = _pack_b_groups      # grouped B streams: one global_load_dwordx4 per 4 MFMAs (layout in k7_blend.hip)
            N = _pack_b16           # narrow layers (<= 16 outputs): 16x16x4 tiles, reduction groups (k0, S)
            self.tensors = [N(rd1[0], [(0, 2)]), _pad32(rd1[1]), P(rd2[0]), _pad32(rd2[1]), P(b1[0]), _pad32(b1[1]), P(b2[0]), _pad32(b2[1]),
                            P(v1[0]), _pad32(v1[1]), P(v2[0][:32]), _pad32(v2[1][:32]), _c(v2[0][32].clone()),
                            P(u1[0]), _pad32(u1[1]), _c(u2[0][0].clone()),
                            N(r1[0], [(0, 4), (16, 4), (32, 2)]), _pad32(r1[1]), N(r2[0], [(0, 4)]), _pad32(r2[1]), _c(r3[0][0].clone())]
            self.scalars = (C.c_float * 4)(float(v2[1][32]), float(u2[1][0]), float(r3[1][0]), float(net.s.detach().abs()))
            self.finite = bool(torch.stack([torch.isfinite(p.detach()).all() for p in net.parameters()]).all())
            self.t_stream, self.t_tab = _pack_blend_t(dict(rd1=rd1, rd2=rd2, b1=b1, b2=b2, v1=v1, v2=v2, u1=u1, u2=u2, r1=r1, r2=r2, r3=r3),
                                                      self.n_feat)
            assert self.t_stream.shape[0] == L.load().gens_blend_views_t_groups((self.n_feat - 3) // 4) + 2
        self.table = L.ptr_table(self.tensors)
        self.key = BlendPlan.version(net)


def blend_views(plan, views, pts, index=None, rgb_out=None, vis_out=None, count=None):
    """Blended colour of pts[index] (N,3) and the per-source in-frustum flags (N,S) written at index (dense outputs)."""
    pts = _c(pts.detach().reshape(-1, 3).to(_f32))
    n = pts.shape[0] if index is None else index.shape[0]
    s = views.nv - 1
    nl = len(views.feat_tex)
    assert plan.n_feat == 3 + 4 * nl, "colour network width does not match the feature pyramid"
    if rgb_out is None:
        rgb_out = torch.zeros(pts.shape[0], 3, device=pts.device, dtype=_f32)
    if vis_out is None:
        vis_out = torch.zeros(pts.shape[0], s, device=pts.device, dtype=torch.uint8)
    idx = None if index is None else _c(index.to(torch.int64))
    hw = [d for f in views.feat_tex for d in f.shape[1:3]]
    feats = [aligned16(f.detach()) for f in views.feat_tex]
    f = plan.n_feat
    flops = 2 * s * (4 * 16 + 16 * f + 3 * f * 64 + 64 * 32 + 32 * 32 + 32 * 33 + 32 * 32 + 32 + 37 * 16 + 16 * 8 + 8)
    nbytes = n * (12 + 12 + s + (8 if idx is not None else 0))
    if 2 <= s <= 4 and kernels.blend == "transposed":                  # two to four source views: the transposed kernel (k7t_blend.hip)
        L.call("gens_blend_views_t", L.ptr_table(feats, align=16), L.int_table(hw), nl, L.ptr(aligned16(views.imgs_tex.detach()), align=16),
               L.ptr(views.w2c), L.ptr(views.intr), L.ptr(views.c2w), views.nv, L.ptr(plan.t_stream), L.ptr(plan.t_tab), plan.scalars, L.ptr(pts),
               L.ptr(idx, torch.int64), n, L.ptr(count, torch.int32), L.ptr(rgb_out), L.ptr(vis_out, torch.uint8), L.stream(),
               live=None if count is None else (count, n), nbytes=nbytes, flops=n * flops, label="gens_blend_views")
        if not plan.finite:
            _poison(idx, count, rgb_out)
        return rgb_out, vis_out
    L.call("gens_blend_views", L.ptr_table(feats, align=16), L.int_table(hw), nl, L.ptr(aligned16(views.imgs_tex.detach()), align=16), L.ptr(views.w2c), L.ptr(views.intr),
           L.ptr(views.c2w), views.nv, plan.table, plan.scalars, L.ptr(pts), L.ptr(idx, torch.int64), n, L.ptr(count, torch.int32),
           L.ptr(rgb_out), L.ptr(vis_out, torch.uint8), L.stream(), live=None if count is None else (count, n), nbytes=nbytes, flops=n * flops)
    if not plan.finite:
        _poison(idx, count, rgb_out)
    return rgb_out, vis_out


# ------------------------------------------------------------------------------------------------------------------
# K18  lookup_feature + BlendingNetwork of a training step, forward and backward   (projector.py:278-349, blending_network.py:69-118)
# ------------------------------------------------------------------------------------------------------------------
def blend_params(net):
    """The 23 raw parameters of a BlendingNetwork in the order gens_blend_train_* read them."""
    mods = [net.ray_dir_fc[0], net.ray_dir_fc[2], net.base_fc[0], net.base_fc[2], net.vis_fc[0], net.vis_fc[2], net.vis_fc2[0], net.vis_fc2[2],
            net.rgb_fc[0], net.rgb_fc[2], net.rgb_fc[4]]
    out = []
    for m in mods:
        out += [m.weight, m.bias]
    return out + [net.s]


class _BlendTrain(torch.autograd.Function):
    """(pts, views, *23 parameters, imgs_tex, *feat_tex) -> (rgb (N,3), vis (N,S) uint8).  Backward: one launch that recomputes the forward
    of its rows and walks the layers in reverse, one batched K14 launch for the eleven [dW | db], K4's backward for the maps."""

    @staticmethod
    def forward(ctx, pts, views, sel, *tensors):
        params, imgs_tex, feat_tex = tensors[:23], tensors[23], tensors[24:]
        # sel (StepPoints): the ray samples among sel.idx[:count_ray] (the list is sorted, ray samples first), dense outputs in sel
        n, s, nl = (pts.shape[0] if sel is None else sel.n_ray), views.nv - 1, len(feat_tex)
        dev = pts.device
        idx, cnt = (None, None) if sel is None else (sel.idx, sel.counts[1:2])
        w = [_c(p.detach().to(_f32)).reshape(-1) if p.dim() == 0 else _c(p.detach().to(_f32)) for p in params]
        feats = [aligned16(f.detach()) for f in feat_tex]
        imgs = aligned16(imgs_tex.detach())
        hw = [d for f in feats for d in f.shape[1:3]]
        if sel is None:
            rgb = torch.empty(n, 3, device=dev, dtype=_f32)
            vis = torch.empty(n, s, device=dev, dtype=torch.uint8)
        else:
            rgb, vis = sel.rgb, sel.vis
        f = 3 + 4 * nl
        flops = 2 * s * (4 * 16 + 16 * f + 3 * f * 64 + 64 * 32 + 32 * 32 + 32 * 33 + 32 * 32 + 32 + 37 * 16 + 16 * 8 + 8)
        ctx.args = (L.ptr_table(feats, align=16), L.int_table(hw), nl, L.ptr(imgs, align=16), L.ptr(views.w2c), L.ptr(views.intr), L.ptr(views.c2w),
                    views.nv, L.ptr_table(w), L.ptr(pts), L.ptr(idx, torch.int64), n, L.ptr(cnt, torch.int32))
        ctx.keep = (feats, imgs, w, views, pts, hw, idx, cnt)
        ctx.meta = (n, s, nl, f, flops, [p.shape for p in params], [t.shape for t in feat_tex], imgs_tex.shape)
        ctx.live = None if cnt is None else (cnt, n)
        if 2 <= s <= 4 and kernels.blend_train_fwd == "transposed":
            # the forward values come from the TRANSPOSED inference kernel (k7t_blend.hip: 5 - 6 x the rate of the row-major training kernel),
            # its weight stream packed from this step's raw parameters by one launch; the backward launch recomputes what it differentiates
            groups = L.load().gens_blend_views_t_groups(nl)
            wstream = torch.empty((groups + 2) * 64 * 4, device=dev, dtype=_f32)
            tab, sc = torch.empty(320, device=dev, dtype=_f32), torch.empty(4, device=dev, dtype=_f32)
            L.call("gens_blend_pack_t", L.ptr_table(w), nl, L.ptr(wstream, align=16), L.ptr(tab), L.ptr(sc), L.stream())
            L.call("gens_blend_views_t_dev", ctx.args[0], ctx.args[1], nl, ctx.args[3], ctx.args[4], ctx.args[5], ctx.args[6], views.nv, L.ptr(wstream),
                   L.ptr(tab), L.ptr(sc), L.ptr(pts), L.ptr(idx, torch.int64), n, L.ptr(cnt, torch.int32), L.ptr(rgb), L.ptr(vis, torch.uint8), L.stream(),
                   nbytes=n * (24 + s), flops=n * flops, live=ctx.live, label="gens_blend_train_fwd")
        else:
            L.call("gens_blend_train_fwd", *ctx.args, L.ptr(rgb), L.ptr(vis, torch.uint8), L.stream(), nbytes=n * (24 + s), flops=n * flops, live=ctx.live)
        ctx.mark_non_differentiable(vis)
        return rgb, vis

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_rgb, _g_vis):
        n, s, nl, f, flops, pshapes, fshapes, ishape = ctx.meta
        feats, imgs, w, views, pts, hw, idx, cnt = ctx.keep
        dev = pts.device
        rows = L.load().gens_blend_train_rows(n, views.nv)
        ins = [4, 16, 3 * f, 64, 32, 32, 32, 32, 37, 16, 8]
        outs = [16, f, 64, 32, 32, 33, 32, 1, 16, 8, 1]
        e = lambda *shape: torch.empty(*shape, device=dev, dtype=_f32)  # noqa: E731
        ev = lambda x: (x + 1) // 2 * 2  # noqa: E731      (even widths: the batched product then reads 8 bytes per lane)
        want_maps = any(ctx.needs_input_grad[3 + 23:])
        g_feat = e(n, s, f) if want_maps else None
        true_flops = (2 * flops + 2 * s * sum(m * (k + 1) for m, k in zip(outs, ins))) * n      # the forward again, the reverse chain, [dW | db]
        # (round 5 counted 3 x the forward's products here -- the reverse chain is as long as the forward, not twice: its rate read 4/3 too high.
        #  bench.py prints both; the ratio of the two counts for this shape:)
        kernels.blend_bwd_round5_count_ratio = (3 * flops + 2 * s * sum(m * (k + 1) for m, k in zip(outs, ins))) * n / max(true_flops, 1)
        if 2 <= s <= 4 and kernels.blend_train_bwd == "transposed" and kernels.blend_train_wgrad == "inside" and n > 0:
            # the TRANSPOSED backward (k18t_blend_train.hip): a wave per 16 rows, the weights in LDS, one block of weight-gradient sums per wave
            lib = L.load()
            csz = lib.gens_blend_train_acc_floats(nl)
            n_parts = lib.gens_blend_train_t_parts(n, views.nv)
            parts, cc, s_part = e(n_parts, csz), e(csz), e(n_parts)
            L.call("gens_blend_train_bwd_t", *ctx.args, L.ptr(_c(g_rgb.to(_f32))), L.ptr(g_feat), L.ptr(s_part), L.ptr(parts), L.ptr(cc), L.stream(),
                   nbytes=4 * n * (3 + s * f), flops=true_flops, live=ctx.live, label="gens_blend_train_bwd")
            return _BlendTrain._finish(ctx, cc, s_part, g_feat, w, f, nl, pshapes, fshapes, ishape, want_maps, views, pts, hw, idx, cnt, n, dev)
        s_part = e(rows // 32) if n else torch.zeros(rows // 32, device=dev, dtype=_f32)
        if kernels.blend_train_wgrad == "inside":
            # the eleven [dW_l | db_l] blocks summed INSIDE the backward launch (persistent workgroups, sums in registers): no operand rows
            lib = L.load()
            csz = lib.gens_blend_train_acc_floats(nl)
            assert csz == sum(ev(m) * ev(k + 1) for m, k in zip(outs, ins))
            parts = e(lib.gens_blend_train_acc_parts(n, views.nv), csz)
            # (zeros, not empty: with n == 0 the entry point returns before either of its launches and _finish reads cc as it is)
            cc = torch.zeros(csz, device=dev, dtype=_f32) if n == 0 else e(csz)
            L.call("gens_blend_train_bwd_acc", *ctx.args, L.ptr(_c(g_rgb.to(_f32))), L.ptr(g_feat), L.ptr(s_part), L.ptr(parts), L.ptr(cc), L.stream(),
                   nbytes=4 * n * (3 + s * f), flops=true_flops, live=ctx.live, label="gens_blend_train_bwd")
            return _BlendTrain._finish(ctx, cc, s_part, g_feat, w, f, nl, pshapes, fshapes, ishape, want_maps, views, pts, hw, idx, cnt, n, dev)
        r_ops = [e(rows, ev(k + 1)) for k in ins]
        l_ops = [e(rows, ev(m)) for m in outs]
        L.call("gens_blend_train_bwd", *ctx.args, L.ptr(_c(g_rgb.to(_f32))), L.ptr_table(r_ops), L.ptr_table(l_ops), L.ptr(g_feat), L.ptr(s_part),
               L.stream(), nbytes=4 * rows * (sum(ins) + 11 + sum(outs)), flops=2 * n * flops, live=ctx.live)      # the forward again + the reverse chain
        # [dW_l | db_l] = l_ops[l]^T r_ops[l]: eleven products over the same rows in one launch
        ms, ns = [ev(m) for m in outs], [ev(k + 1) for k in ins]
        mi, ni = L.int_table(ms), L.int_table(ns)
        ws = e(L.load().gens_gemm_tn_batch_workspace(11, mi, ni, rows))
        cc = e(sum(m * k for m, k in zip(ms, ns)))
        if cnt is None:
            L.call("gens_gemm_tn_batch", 11, L.ptr_table(l_ops), mi, L.ptr_table(r_ops), ni, mi, ni, rows, L.ptr(ws), L.ptr(cc), L.stream(),
                   nbytes=4 * rows * (sum(ms) + sum(ns)), flops=2 * rows * sum(m * k for m, k in zip(ms, ns)))
        else:       # floor(32 / S) points -> 32 operand rows per workgroup of the backward launch
            L.call("gens_gemm_tn_batch_live", 11, L.ptr_table(l_ops), mi, L.ptr_table(r_ops), ni, mi, ni, rows, L.ptr(cnt, torch.int32), 32 // s, 32,
                   L.ptr(ws), L.ptr(cc), L.stream(), nbytes=4 * rows * (sum(ms) + sum(ns)), flops=2 * rows * sum(m * k for m, k in zip(ms, ns)),
                   live=ctx.live, label="gens_gemm_tn_batch")
        return _BlendTrain._finish(ctx, cc, s_part, g_feat, w, f, nl, pshapes, fshapes, ishape, want_maps, views, pts, hw, idx, cnt, n, dev)

    @staticmethod
    def _finish(ctx, cc, s_part, g_feat, w, f, nl, pshapes, fshapes, ishape, want_maps, views, pts, hw, idx, cnt, n, dev):
        e = lambda *shape: torch.empty(*shape, device=dev, dtype=_f32)  # noqa: E731
        # the 23 parameter gradients out of the product blocks in one launch, as views of one flat buffer (contiguous each)
        sizes = [math.prod(sh) if len(sh) else 1 for sh in pshapes]
        flat = e(sum(sizes))
        grads, off = [], 0
        for sh, sz in zip(pshapes, sizes):
            grads.append(flat[off:off + sz].view(sh))
            off += sz
        L.call("gens_blend_train_wgrad", L.ptr(cc), L.ptr(s_part), s_part.numel(), L.ptr(w[22]), f, L.ptr_table([g_.reshape(-1) for g_ in grads]), L.stream())
        g_imgs, g_feats = None, [None] * nl
        if want_maps:
            want_img = ctx.needs_input_grad[3 + 23]
            want_feat = any(ctx.needs_input_grad[3 + 24:])
            g_feats_t = [torch.zeros(sh, device=dev, dtype=_f32) for sh in fshapes] if want_feat else None
            g_imgs = torch.zeros(ishape, device=dev, dtype=_f32) if want_img else None
            L.call("gens_lookup_feature_bwd_idx", L.int_table(hw), nl, L.ptr(views.w2c), L.ptr(views.intr), views.nv, L.ptr(pts), L.ptr(g_feat),
                   L.ptr(idx, torch.int64), n, L.ptr(cnt, torch.int32), L.ptr_table(g_feats_t), L.ptr(g_imgs), L.stream(), label="gens_lookup_feature_bwd")
            if want_feat:
                g_feats = g_feats_t
        return (None, None, None, *grads, g_imgs, *g_feats)


def blend_train(net, views, pts, sel=None):
    """Colour of every point blended from the source views, differentiable with respect to the network and the maps:
    -> (rgb (N,3), vis (N,S) bool).  pts (N,3) device float32 (no gradient flows to the points, as in the reference's call).
    sel (StepPoints): only the selected ray samples are evaluated (count on the device); rgb / vis are sel's dense arrays."""
    pts = _c(pts.detach().reshape(-1, 3).to(_f32))
    rgb, vis = _BlendTrain.apply(pts, views, sel, *blend_params(net), views.imgs_tex, *views.feat_tex)
    return rgb, (vis if sel is not None else vis.bool())        # (a selection's flags stay uint8: the compositing launch reads them as they are)


class StepPoints:
    """The masked evaluation set of ONE training render (implicit_surface.py:174-191,256-257,484-497) with nothing read back to the host:
    dense rows [ray samples | always-evaluated points | pseudo points] in one (N, 3) buffer, the selected rows as a device-side index list +
    counts (gens_compact_points: the reference's nonzero + first-ten rescue), and the dense outputs of the two networks, whose unselected
    rows the same launch fills with the reference's defaults (Q8).  The same launch computes max(z_vals) (:301) and
    inv_s = clip(exp(10 variance), 1e-6, 1e6) (:206) into `scalars`."""

    def __init__(self, pts_all, valid_all, n_ray, n_always, n_src, z=None, variance=None):
        dev = pts_all.device
        self.pts = pts_all
        self.n, self.n_ray, self.n_always = int(pts_all.shape[0]), int(n_ray), int(n_always)
        n = self.n
        self.idx = torch.empty(n, device=dev, dtype=torch.int64)
        self.counts = torch.empty(3, device=dev, dtype=torch.int32)
        self.y, self.g, self.s = (torch.empty(n, k, device=dev, dtype=_f32) for k in (1, 3, 3))
        self.rgb = torch.empty(self.n_ray, 3, device=dev, dtype=_f32)
        self.vis = torch.empty(self.n_ray, n_src, device=dev, dtype=torch.uint8)
        self.scalars = torch.empty(4, device=dev, dtype=_f32)
        zc = None if z is None else _c(z.detach())
        scratch = torch.empty(L.load().gens_compact_points_scratch(n), device=dev, dtype=torch.int32)
        L.call("gens_compact_points", L.ptr(_c(valid_all), torch.uint8), self.n_ray, self.n_always, n, L.ptr(self.idx, torch.int64),
               L.ptr(self.counts, torch.int32), L.ptr(self.y), L.ptr(self.g), L.ptr(self.s), L.ptr(self.rgb), L.ptr(self.vis, torch.uint8), n_src,
               L.ptr(zc), 0 if zc is None else zc.numel(), L.ptr(None if variance is None else _c(variance.detach().reshape(1))), L.ptr(self.scalars),
               L.ptr(scratch, torch.int32), L.stream(), nbytes=n * 9)


__all__ = [n_ for n_ in dir() if not n_.startswith("__")]      # private helpers travel too: the package namespace is the old module's
