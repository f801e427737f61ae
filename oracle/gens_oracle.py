"""CPU oracle for the GenS hot path -- TEST INFRASTRUCTURE, never the product path.

A restatement (float32, torch on CPU, written point-by-point rather than as the reference's
tensor pipeline) of what the reference computes for each kernel of SURVEY.md §8(a).  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package; nothing under ``gens_amd/`` does.

Pinning: every function below is checked against golden vectors produced by the reference's own
Python (``tests/golden/make_golden.py``; ``tests/test_oracle_golden.py``).  The reference has no
tests or known-answer vectors of its own, and its only native kernel (the CUDA
second-order sampler) cannot run here; second-order goldens come from the reference's pure-torch
sampler (projector.py:62-214) for in-cube points and from a zeros-padding stand-in validated
against it (see make_golden.py).  Citations are relative to /root/reference.

Axis convention (Q1): a volume tensor is (1, C, X, Y, Z); a world point p = (px, py, pz) indexes
[c, ix(px), iy(py), iz(pz)].  The reference reaches the same element by flipping the point and
letting grid_sample's (x->W, y->H, z->D) convention undo the flip (projector.py:223).
"""
import math

import torch

F32 = torch.float32


# ----------------------------------------------------------------------------------------------
# small helpers
# ----------------------------------------------------------------------------------------------
def _bilinear(img, ix, iy):
    """Zeros-padded bilinear read of img (C,H,W) at float pixel coords; differentiable in ix, iy, img.

    Corner weights are formed like ATen's grid sampler ((x_se - x) * (y_se - y) ...), see
    torch/include/ATen/native/cuda/GridSampler.cuh.
    """
    c, h, w = img.shape
    fin = torch.isfinite(ix) & torch.isfinite(iy)
    sx = torch.where(fin, ix, torch.zeros_like(ix))
    sy = torch.where(fin, iy, torch.zeros_like(iy))
    x0 = torch.floor(sx.detach()).clamp(-4, w + 4)
    y0 = torch.floor(sy.detach()).clamp(-4, h + 4)
    flat = img.reshape(c, h * w)
    out = 0
    for dy in (0, 1):
        for dx in (0, 1):
            xi, yi = x0 + dx, y0 + dy
            wx = (sx - x0) if dx else (x0 + 1 - sx)
            wy = (sy - y0) if dy else (y0 + 1 - sy)
            ok = fin & (xi >= 0) & (xi < w) & (yi >= 0) & (yi < h)
            lin = (yi.clamp(0, h - 1) * w + xi.clamp(0, w - 1)).long()
            out = out + flat[:, lin] * (wx * wy * ok.to(img.dtype))[None]
    return out  # (C, N)


def _trilinear(vol, pts):
    """vol (C,X,Y,Z), pts (N,3) in [-1,1] (align_corners=True, zeros padding) -> (N,C); differentiable."""
    c, dx_, dy_, dz_ = vol.shape
    size = torch.tensor([dx_, dy_, dz_], dtype=pts.dtype)
    pos = (pts + 1) * 0.5 * (size - 1)
    base = torch.floor(pos.detach())
    flat = vol.reshape(c, -1)
    out = 0
    for ox in (0, 1):
        for oy in (0, 1):
            for oz in (0, 1):
                off = torch.tensor([ox, oy, oz], dtype=pts.dtype)
                idx = base + off
                w = torch.where(off > 0, pos - base, base + 1 - pos)
                ok = ((idx >= 0) & (idx < size)).all(-1)
                ii = torch.minimum(idx.clamp(min=0), size - 1).long()
                lin = (ii[:, 0] * dy_ + ii[:, 1]) * dz_ + ii[:, 2]
                out = out + flat[:, lin] * (w[:, 0] * w[:, 1] * w[:, 2] * ok.to(vol.dtype))[None]
    return out.t()


# ----------------------------------------------------------------------------------------------
# K1  Volume.agg_mean_var  (models/modules/volume.py:13-63)
# ----------------------------------------------------------------------------------------------
def volume_build(features, intrs, c2ws, dims, min_vis_view=1, x_ranges=None):
    """-> (volumes [(1,2C,D,D,D)], masks [(1,1,D,D,D)]).  Differentiable w.r.t. features only.

    x_ranges (tests at BASELINE's full volume sizes): per level an index range (x0, x1) along the SLOWEST volume axis (world x, Q1) -- only
    that slab of voxels is built, -> (1,2C,x1-x0,D,D) / (1,1,x1-x0,D,D); voxels are independent (volume.py:27-61), so the slab is the
    corresponding slice of the whole cube (tests/test_oracle_golden.py checks that on the golden sizes)."""
    w2cs = torch.inverse(c2ws)
    vols, masks = [], []
    for lvl, d in enumerate(dims):
        x0, x1 = (0, d) if x_ranges is None or x_ranges[lvl] is None else x_ranges[lvl]
        n_slab = x1 - x0
        feat = features[lvl]
        nv, c, h, w = feat.shape
        k = intrs.clone()
        k[:, :2] = k[:, :2] * 0.5 ** lvl                                   # volume.py:24-25 (Q2)
        g = torch.linspace(-1, 1, d, dtype=F32)
        xs, ys, zs = torch.meshgrid(g[x0:x1], g, g, indexing="ij")        # (Q1) x is the slowest axis
        world = torch.stack([xs.reshape(-1), ys.reshape(-1), zs.reshape(-1), torch.ones(n_slab * d * d)], 0)
        s1 = torch.zeros(c, n_slab * d * d)
        s2 = torch.zeros(c, n_slab * d * d)
        cnt = torch.zeros(n_slab * d * d)
        for v in range(nv):
            with torch.no_grad():
                cam = w2cs[v] @ world
                img = (k[v] @ cam)[:3]
                px = img[0] / (img[2] + 1e-8)                            # volume.py:36 (Q3)
                py = img[1] / (img[2] + 1e-8)
                nx = px / ((w - 1) / 2) - 1
                ny = py / ((h - 1) / 2) - 1
                vis = ((nx.abs() <= 1) & (ny.abs() <= 1) & (img[2] > 0)).to(F32)
                ix = (nx + 1) / 2 * (w - 1)                              # align_corners=True, volume.py:46
                iy = (ny + 1) / 2 * (h - 1)
            fw = _bilinear(feat[v], ix, iy) * vis[None]
            s1 = s1 + fw
            s2 = s2 + fw ** 2
            cnt = cnt + vis
        den = torch.where(cnt <= 0, torch.full_like(cnt, 1e-8), cnt)      # volume.py:53 (Q5)
        mean = s1 / den
        var = s2 / den - mean ** 2
        vols.append(torch.cat([mean, var], 0).reshape(1, 2 * c, n_slab, d, d))
        masks.append((cnt > min_vis_view).to(F32).reshape(1, 1, n_slab, d, d))  # (Q4)
    return vols, masks


# ----------------------------------------------------------------------------------------------
# K2 / K2''  lookup_volume(..., "grad")  (projector.py:217-245, cuda_gridsample.py:71-123,
#            gridsample_cuda.cu:212-533)
# ----------------------------------------------------------------------------------------------
def lookup_volume(volumes, pts):
    """volumes: list of (1,C,X,Y,Z); pts (N,3) -> (N, sum C).  Differentiable to any order."""
    return torch.cat([_trilinear(v[0], pts) for v in volumes], -1)


def lookup_volume_bwd(g_out, volumes, pts):
    """First backward: -> ([gV_l], gP).  What aten::grid_sampler_3d_backward returns per level, summed over levels for gP."""
    vols = [v.detach().requires_grad_(True) for v in volumes]
    p = pts.detach().requires_grad_(True)
    grads = torch.autograd.grad(lookup_volume(vols, p), vols + [p], g_out)
    return list(grads[:-1]), grads[-1]


def lookup_volume_bwd2(gg_vols, gg_pts, g_out, volumes, pts):
    """Backward of the backward (grad2_3d, gridsample_cuda.cpp:42-56): -> (ggO, [gV'_l], gP')."""
    vols = [v.detach().requires_grad_(True) for v in volumes]
    p = pts.detach().requires_grad_(True)
    go = g_out.detach().requires_grad_(True)
    grads = torch.autograd.grad(lookup_volume(vols, p), vols + [p], go, create_graph=True)
    phi = (grads[-1] * gg_pts).sum()
    if gg_vols is not None:
        phi = phi + sum((a * b).sum() for a, b in zip(grads[:-1], gg_vols))
    outs = torch.autograd.grad(phi, [go] + vols + [p], allow_unused=True)
    outs = [torch.zeros_like(t) if o is None else o for o, t in zip(outs, [go] + vols + [p])]
    return outs[0], list(outs[1:-1]), outs[-1]


class _TruncatedLookup(torch.autograd.Function):
    """lookup_volume as the reference's Function pair sees it: differentiable twice, third order dropped
    (the outputs of grad2_3d are plain tensors, cuda_gridsample.py:110-123)."""

    @staticmethod
    def forward(ctx, pts, *volumes):
        ctx.save_for_backward(pts, *volumes)
        with torch.no_grad():
            return lookup_volume(list(volumes), pts)

    @staticmethod
    def backward(ctx, g_out):
        pts, *volumes = ctx.saved_tensors
        outs = _TruncatedLookupBwd.apply(g_out, pts, *volumes)
        return (outs[0],) + tuple(outs[1:])


class _TruncatedLookupBwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g_out, pts, *volumes):
        ctx.save_for_backward(g_out, pts, *volumes)
        with torch.enable_grad():
            gv, gp = lookup_volume_bwd(g_out, list(volumes), pts)
        return (gp.detach(),) + tuple(g.detach() for g in gv)

    @staticmethod
    def backward(ctx, gg_pts, *gg_vols):
        g_out, pts, *volumes = ctx.saved_tensors
        ggv = None if all(g is None for g in gg_vols) else [torch.zeros_like(v) if g is None else g for g, v in zip(gg_vols, volumes)]
        with torch.enable_grad():
            ggo, gv2, gp2 = lookup_volume_bwd2(ggv, gg_pts, g_out, list(volumes), pts)
        return (ggo.detach(), gp2.detach()) + tuple(g.detach() for g in gv2)


def lookup_volume_truncated(volumes, pts):
    return _TruncatedLookup.apply(pts, *volumes)


# ----------------------------------------------------------------------------------------------
# K3  lookup_volume(..., "nearest")  (projector.py:231,240)  -- align_corners=False (Q6)
# ----------------------------------------------------------------------------------------------
def lookup_mask_nearest(masks, pts):
    """-> (N, L) float values of the nearest voxel (0 outside)."""
    cols = []
    for m in masks:
        vol = m[0, 0]
        size = torch.tensor(vol.shape, dtype=pts.dtype)
        idx = torch.round(((pts + 1) * size - 1) / 2)                      # torch.round == nearbyint (half-to-even)
        ok = ((idx >= 0) & (idx < size)).all(-1)
        ii = torch.minimum(idx.clamp(min=0), size - 1).long()
        cols.append(vol[ii[:, 0], ii[:, 1], ii[:, 2]] * ok.to(vol.dtype))
    return torch.stack(cols, -1)


def point_valid(masks, pts):
    """(N,) bool: any level's nearest mask voxel is set (Q7, without the 'first 10' rescue)."""
    return (lookup_mask_nearest(masks, pts) > 0).any(-1)


# ----------------------------------------------------------------------------------------------
# K4  lookup_feature + compute_angle  (projector.py:278-349)
# ----------------------------------------------------------------------------------------------
def lookup_feature(pts, imgs, intrs, c2ws, features):
    """-> feat_views (N,S,3+sum C), ray_diff (N,S,4), mask (N,S) bool.  Differentiable in imgs/features."""
    n = pts.shape[0]
    s = intrs.shape[0] - 1
    w2cs = torch.inverse(c2ws[1:])
    ph = torch.cat([pts.t(), torch.ones(1, n)], 0)
    rgb, per_level, mask = None, [], torch.ones(n, s, dtype=torch.bool)
    for lvl, feat in enumerate(features):
        _, c, h, w = feat.shape
        k = intrs[1:, :3, :3].clone()
        k[:, :2] = k[:, :2] * 0.5 ** lvl
        cols = []
        rgb_cols = []
        for v in range(s):
            with torch.no_grad():
                cam = k[v] @ (w2cs[v] @ ph)[:3]
                px = cam[0] / cam[2]                                     # no epsilon here (Q3)
                py = cam[1] / cam[2]
                nx = px / ((w - 1) / 2) - 1
                ny = py / ((h - 1) / 2) - 1
                mask[:, v] &= (cam[2] > 0) & (px >= 0) & (px < w) & (py >= 0) & (py < h)
                ix = ((nx + 1) * w - 1) / 2                              # align_corners=False read (Q6)
                iy = ((ny + 1) * h - 1) / 2
            cols.append(_bilinear(feat[v + 1], ix, iy).t())
            if lvl == 0:
                rgb_cols.append(_bilinear(imgs[v + 1], ix, iy).t())
        per_level.append(torch.stack(cols, 1))
        if lvl == 0:
            rgb = torch.stack(rgb_cols, 1)
    # compute_angle, projector.py:278-291
    to_ref = c2ws[0, :3, 3][None, None] - pts[:, None]
    to_ref = to_ref / (torch.linalg.norm(to_ref, dim=-1, keepdim=True) + 1e-6)
    to_src = c2ws[1:, :3, 3][None] - pts[:, None]
    to_src = to_src / (torch.linalg.norm(to_src, dim=-1, keepdim=True) + 1e-6)
    diff = to_ref - to_src
    dn = torch.linalg.norm(diff, dim=-1, keepdim=True)
    dot = (to_ref * to_src).sum(-1, keepdim=True)
    ray_diff = torch.cat([diff / dn.clamp(min=1e-6), dot], -1)
    return torch.cat([rgb] + per_level, -1), ray_diff, mask


# ----------------------------------------------------------------------------------------------
# K5-K7  up_sample / sample_pdf / cat_z_vals  (implicit_surface.py:14-44, 60-133)
# ----------------------------------------------------------------------------------------------
def sample_pdf_det(bins, weights, n_new):
    """Deterministic inverse-CDF sampling, quantiles (k+.5)/n_new."""
    b, nb = bins.shape
    w = weights + 1e-5
    pdf = w / w.sum(-1, keepdim=True)
    cdf = torch.cat([torch.zeros(b, 1), torch.cumsum(pdf, -1)], -1)
    u = torch.linspace(0.5 / n_new, 1 - 0.5 / n_new, n_new, dtype=F32)
    out = torch.empty(b, n_new)
    for r in range(b):
        for j in range(n_new):
            above = int((cdf[r] <= u[j]).sum())                            # searchsorted(right=True)
            lo = max(above - 1, 0)
            hi = min(above, nb - 1)
            den = cdf[r, hi] - cdf[r, lo]
            den = den if den >= 1e-5 else torch.tensor(1.0)
            t = (u[j] - cdf[r, lo]) / den
            out[r, j] = bins[r, lo] + t * (bins[r, hi] - bins[r, lo])
    return out


def up_sample(rays_o, rays_d, z, sdf, n_new, masks, inv_s):
    """One importance round: -> (B, n_new) new depths (implicit_surface.py:60-109)."""
    b, n = z.shape
    pts = rays_o[:, None] + rays_d[:, None] * z[..., None]
    vm = point_valid(masks, pts.reshape(-1, 3)).reshape(b, n).to(F32)
    vm = vm[:, :-1] * vm[:, 1:]
    rad = torch.linalg.norm(pts, dim=-1)
    inside = ((rad[:, :-1] < 1.0) | (rad[:, 1:] < 1.0)) & (vm > 0)
    mid = (sdf[:, :-1] + sdf[:, 1:]) * 0.5
    dz = z[:, 1:] - z[:, :-1]
    cos = (sdf[:, 1:] - sdf[:, :-1]) / (dz + 1e-5)
    prev = torch.cat([torch.zeros(b, 1), cos[:, :-1]], -1)
    cos = torch.minimum(prev, cos).clamp(-1e3, 0.0) * inside
    e_prev = mid - cos * dz * 0.5
    e_next = mid + cos * dz * 0.5
    c_prev = torch.sigmoid(e_prev * inv_s)
    c_next = torch.sigmoid(e_next * inv_s)
    alpha = (c_prev - c_next + 1e-5) / (c_prev + 1e-5)
    trans = torch.cumprod(torch.cat([torch.ones(b, 1), 1 - alpha + 1e-7], -1), -1)[:, :-1]
    return sample_pdf_det(z, alpha * trans, n_new)


def merge_samples(z, z_new, sdf=None, sdf_new=None):
    """cat + sort (implicit_surface.py:115-116, 127-131).  Ties keep the older sample first."""
    zc = torch.cat([z, z_new], -1)
    zs, idx = torch.sort(zc, dim=-1, stable=True)
    if sdf is None:
        return zs, None
    return zs, torch.gather(torch.cat([sdf, sdf_new], -1), 1, idx)


# ----------------------------------------------------------------------------------------------
# K8  render_core compositing  (implicit_surface.py:160-168, 202-303)
# ----------------------------------------------------------------------------------------------
def composite(rays_o, rays_d, z, sample_dist, sdf, gradients, smooth, color, voxel_mask, src_vis, inv_s,
              cos_anneal, c2w_ref):
    """All per-ray outputs of render_core after the networks have run.  Differentiable torch.

    sdf (B,n) [100 where invalid], gradients/smooth/color (B,n,3) [0 where invalid], voxel_mask (B,n) float,
    src_vis (B,n,S) bool, inv_s scalar tensor (already clipped).
    """
    b, n = z.shape
    dists = torch.cat([z[:, 1:] - z[:, :-1], torch.full((b, 1), sample_dist)], -1)   # (Q10)
    mid_z = z + dists * 0.5
    pts = rays_o[:, None] + rays_d[:, None] * mid_z[..., None]
    valid_mask = ((src_vis.to(F32).sum(-1) > 1).to(F32).sum(-1, keepdim=True) > 8)       # (Q12)

    true_cos = (rays_d[:, None] * gradients).sum(-1)
    iter_cos = -(torch.relu(-true_cos * 0.5 + 0.5) * (1.0 - cos_anneal) + torch.relu(-true_cos) * cos_anneal)
    iter_cos = (iter_cos * voxel_mask).clamp(-10.0, 10.0)
    e_next = sdf + iter_cos * dists * 0.5
    e_prev = sdf - iter_cos * dists * 0.5
    c_prev = torch.sigmoid(e_prev * inv_s)
    c_next = torch.sigmoid(e_next * inv_s)
    alpha = ((c_prev - c_next + 1e-5) / (c_prev + 1e-5)).clamp(0.0, 1.0) * voxel_mask   # (Q9)
    norm = torch.linalg.norm(pts, dim=-1)
    inside = (norm < 1.0).to(F32) * voxel_mask
    relax = (norm < 1.2).to(F32) * voxel_mask
    weights = alpha * torch.cumprod(torch.cat([torch.ones(b, 1), 1 - alpha + 1e-7], -1), -1)[:, :-1]
    wsum = weights.sum(-1, keepdim=True)
    out_color = (color * weights[..., None]).sum(1)
    rot = torch.inverse(c2w_ref[:3, :3])
    normal = ((gradients * weights[..., None]).sum(1)[:, None] @ rot.t()[None])[:, 0]
    cam_d = (rays_d[:, None] @ rot.t()[None])[:, 0]
    render_depth = (mid_z * weights).sum(-1) * cam_d[:, 2]                               # (Q11)

    gnorm = torch.linalg.norm(gradients, dim=-1)
    gradient_error = (relax * (gnorm - 1.0) ** 2).sum() / (relax.sum() + 1e-5)
    sm = (smooth * weights.detach()[..., None] * inside[..., None]).sum(1)
    smooth_error = torch.linalg.norm(sm, dim=-1).abs().mean()

    # first masked sign change, implicit_surface.py:262-299
    pair_ok = ((voxel_mask[:, :-1] * voxel_mask[:, 1:]) > 0).to(F32)
    sign = (sdf[:, :-1] * sdf[:, 1:] <= 0).to(F32)
    rank = torch.arange(n - 1, 0, -1, dtype=F32)
    score = sign * rank[None] * pair_ok
    i0 = torch.argmax(score, 1, keepdim=True)
    i1 = i0 + 1
    mid_in = (0.5 * (inside.gather(1, i0) + inside.gather(1, i1)) > 0.5).to(F32)
    mid_in = mid_in * (score.sum(1, keepdim=True) > 0).to(F32)
    gd = gradients.detach()
    g0 = gd.gather(1, i0[..., None].expand(-1, -1, 3))
    g1 = gd.gather(1, i1[..., None].expand(-1, -1, 3))
    cosd = (g0 * g1).sum(-1) / (torch.linalg.norm(g0, dim=-1) * torch.linalg.norm(g1, dim=-1) + 1e-8)
    mid_in = mid_in * (cosd > 0.5)
    s0, s1 = sdf.gather(1, i0), sdf.gather(1, i1)
    z0, z1 = mid_z.gather(1, i0), mid_z.gather(1, i1)
    z_cross = (s0 * z1 - s1 * z0) / (s0 - s1 + 1e-10)
    sdf_depth = z_cross * cam_d[:, 2:3] * mid_in
    z_cross = torch.where(z_cross < 0, torch.zeros_like(z_cross), z_cross)
    z_cross = torch.where(z_cross > z.max(), torch.zeros_like(z_cross), z_cross)
    pts_sdf0 = rays_o[:, None] + rays_d[:, None] * z_cross[..., None]
    return {
        "color_fine": out_color, "render_depth": render_depth, "normal": normal, "weights": weights,
        "weight_sum": wsum, "weight_max": weights.max(-1, keepdim=True)[0], "inside_sphere": inside,
        "valid_mask": valid_mask, "gradient_error": gradient_error, "smooth_error": smooth_error,
        "mid_inside_sphere": mid_in, "sdf_depth": sdf_depth, "pts_sdf0": pts_sdf0, "mid_z": mid_z,
    }


# ----------------------------------------------------------------------------------------------
# K9  surface_patch_warp / patch_homography  (projector.py:353-437)
# ----------------------------------------------------------------------------------------------
def upsample_bilinear_half_pixel(x, h, w):
    """F.interpolate(x, size=(h,w), mode='bilinear') (align_corners=False) restated: x (N,C,hs,ws)."""
    n, c, hs, ws = x.shape
    ys = ((torch.arange(h, dtype=F32) + 0.5) * (hs / h) - 0.5).clamp(min=0)
    xs = ((torch.arange(w, dtype=F32) + 0.5) * (ws / w) - 0.5).clamp(min=0)
    y0 = ys.floor().long().clamp(max=hs - 1)
    x0 = xs.floor().long().clamp(max=ws - 1)
    y1 = (y0 + 1).clamp(max=hs - 1)
    x1 = (x0 + 1).clamp(max=ws - 1)
    ty = (ys - y0)[None, None, :, None]
    tx = (xs - x0)[None, None, None, :]
    top = x[:, :, y0][:, :, :, x0] * (1 - tx) + x[:, :, y0][:, :, :, x1] * tx
    bot = x[:, :, y1][:, :, :, x0] * (1 - tx) + x[:, :, y1][:, :, :, x1] * tx
    return top * (1 - ty) + bot * ty


def patch_warp(pts0, normals, images, intrs, c2ws, patch=11):
    """pts0 (B,1,3) world surface points, normals (B,1,3) unit normals in the reference-camera frame,
    images (nv,C,H,W) -> ref (1,B,P*P,C), src (S,B,P*P,C).  Differentiable in pts0."""
    b = pts0.shape[0]
    nv, c, h, w = images.shape
    s = nv - 1
    half = patch // 2
    r_ref = c2ws[0, :3, :3]
    c_ref = c2ws[0, :3, 3]
    k_ref = intrs[0, :3, :3]
    k_ref_inv = torch.inverse(intrs)[0, :3, :3]
    p = pts0[:, 0]
    x_cam = p @ r_ref + (-(r_ref.t() @ c_ref))[None]
    proj = x_cam @ k_ref.t()
    disp = (normals[:, 0] * x_cam).sum(-1)                  # n . X  (projector.py:372)
    u0 = proj[:, 0] / (proj[:, 2] + 1e-8)
    v0 = proj[:, 1] / (proj[:, 2] + 1e-8)
    offs = torch.arange(-half, half + 1, dtype=F32)
    oy, ox = torch.meshgrid(offs, offs, indexing="ij")      # x fastest inside a patch row
    ox, oy = ox.reshape(-1), oy.reshape(-1)
    up = u0[:, None] + ox[None]
    vp = v0[:, None] + oy[None]
    src_vals = []
    for v in range(1, nv):
        r_src_t = c2ws[v, :3, :3].t()
        rel = r_src_t @ r_ref
        tvec = r_src_t @ (c_ref - c2ws[v, :3, 3])
        hom = rel[None] + (tvec[None, :, None] * normals[:, 0][:, None, :]) / (disp[:, None, None] + 1e-10)
        hom = intrs[v, :3, :3][None] @ hom @ k_ref_inv[None]
        q = torch.stack([up, vp, torch.ones_like(up)], -1) @ hom.transpose(1, 2)
        gx = q[..., 0] / (q[..., 2] + 1e-8)
        gy = q[..., 1] / (q[..., 2] + 1e-8)
        # normalise with (w-1)/2 then read with align_corners=True -> pixel coordinate is unchanged up to rounding
        ix = ((2 * gx / (w - 1) - 1.0) + 1) / 2 * (w - 1)
        iy = ((2 * gy / (h - 1) - 1.0) + 1) / 2 * (h - 1)
        src_vals.append(_bilinear(images[v], ix.reshape(-1), iy.reshape(-1)).t().reshape(b, -1, c))
    ixr = ((2 * up.detach() / (w - 1) - 1.0) + 1) / 2 * (w - 1)
    iyr = ((2 * vp.detach() / (h - 1) - 1.0) + 1) / 2 * (h - 1)
    ref = _bilinear(images[0], ixr.reshape(-1), iyr.reshape(-1)).t().reshape(1, b, -1, c)
    return ref, torch.stack(src_vals, 0)


# ----------------------------------------------------------------------------------------------
# K10  tv_regularization  (implicit_surface.py:135-150)
# ----------------------------------------------------------------------------------------------
def tv_regularization(volumes, masks):
    total = 0
    for lvl, (v, m) in enumerate(zip(volumes, masks)):
        mx = (m[:, :, 1:] * m[:, :, :-1]) > 0
        my = (m[:, :, :, 1:] * m[:, :, :, :-1]) > 0
        mz = (m[..., 1:] * m[..., :-1]) > 0
        den = mx.sum() + 1e-8                                               # (Q13): all three use mx's count
        tx = (((v[:, :, 1:] - v[:, :, :-1]) ** 2) * mx).sum() / den
        ty = (((v[:, :, :, 1:] - v[:, :, :, :-1]) ** 2) * my).sum() / den
        tz = (((v[..., 1:] - v[..., :-1]) ** 2) * mz).sum() / den
        total = total + torch.sqrt(tx + ty + tz) * 0.5 ** lvl
    return total


# ----------------------------------------------------------------------------------------------
# K11  extract_geometry lattice  (implicit_surface.py:407-421)
# ----------------------------------------------------------------------------------------------
def lattice_points(bound_min, bound_max, resolution):
    xs = torch.linspace(float(bound_min[0]), float(bound_max[0]), resolution)
    ys = torch.linspace(float(bound_min[1]), float(bound_max[1]), resolution)
    zs = torch.linspace(float(bound_min[2]), float(bound_max[2]), resolution)
    xx, yy, zz = torch.meshgrid(xs, ys, zs, indexing="ij")
    return torch.stack([xx, yy, zz], -1).reshape(-1, 3)


# ----------------------------------------------------------------------------------------------
# K13  compute_LNCC  (/root/reference/models/losses/ncc.py:7-55)
# ----------------------------------------------------------------------------------------------
def lncc(ref_gray, src_grays):
    """ref_gray (1,B,P,C), src_grays (S,B,P,C) -> (B,1).  The five grouped all-ones convolutions read at the centre tap
    (ncc.py:29-33) are sums over the P = 11x11 patch samples; the rest follows ncc.py:35-53 operation by operation.
    Differentiable (plain torch ops), so the gradients of the golden can be checked through autograd."""
    p = ref_gray.shape[2]
    r = ref_gray.permute(1, 0, 3, 2)            # (B,1,C,P)
    s = src_grays.permute(1, 0, 3, 2)           # (B,S,C,P)
    ref_sum, src_sum = r.sum(-1), s.sum(-1)
    ref_sq_sum, src_sq_sum = (r * r).sum(-1), (s * s).sum(-1)
    ref_src_sum = (r * s).sum(-1)
    u_ref, u_src = ref_sum / p, src_sum / p
    cross = ref_src_sum - u_src * ref_sum - u_ref * src_sum + u_ref * u_src * p
    ref_var = ref_sq_sum - 2 * u_ref * ref_sum + u_ref * u_ref * p
    src_var = src_sq_sum - 2 * u_src * src_sum + u_src * u_src * p
    cc = cross * cross / (ref_var * src_var + 1e-5)
    ncc = torch.clamp(1 - cc, 0.0, 2.0).mean(dim=2)                          # (B,S)
    ncc, _ = torch.topk(ncc, 2, dim=1, largest=False)
    return ncc.mean(dim=1, keepdim=True)
