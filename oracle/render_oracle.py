"""CPU oracle for the render driver (ImplicitSurface.render / extract_geometry) -- TEST INFRASTRUCTURE.

The MLPs are evaluated functionally from a state dict (same parameter names as the reference:
``sdf_network.lin{l}.{weight_g,weight_v,bias}``, ``color_network.*``, ``deviation_network.variance``)
with plain torch autograd on CPU; the gather / per-ray kernels come from ``gens_oracle``.
Pinned by tests/golden/g9*_render.npz and g10_geometry.npz (tests/test_oracle_golden.py).
Citations are relative to /root/reference.
"""
import math

import torch
import torch.nn.functional as F

from . import gens_oracle as K


def _embed(x, n_freq):
    """NeRF positional encoding, include_input, log-sampled bands (embedder.py:6-51)."""
    parts = [x]
    for i in range(n_freq):
        parts += [torch.sin(x * 2.0 ** i), torch.cos(x * 2.0 ** i)]
    return torch.cat(parts, -1)


def _wn_linear(sd, name, x):
    v, g, b = sd[name + ".weight_v"], sd[name + ".weight_g"], sd[name + ".bias"]
    w = v * (g / torch.linalg.norm(v, dim=1, keepdim=True))
    return x @ w.t() + b


def sdf_mlp(sd, pts, volumes, lookup=K.lookup_volume, prefix="sdf_network.", skip_in=(3,), scale=1.0):
    """SDFNetwork.forward (sdf_network.py:98-123): -> (N, 129)."""
    n_lin = len([k for k in sd if k.startswith(prefix + "lin") and k.endswith(".bias")])
    fe = _embed(lookup(volumes, pts), 2)
    xe = _embed(pts * scale, 4)
    h = xe
    for l in range(n_lin):
        if l in skip_in:
            h = torch.cat([h, xe], -1) / math.sqrt(2)
        if l > 0:
            h = torch.cat([h, fe], -1)
        h = _wn_linear(sd, f"{prefix}lin{l}", h)
        if l < n_lin - 1:
            h = F.softplus(h, beta=100)
    return torch.cat([h[:, :1] / scale, h[:, 1:]], -1)


def sdf_gradient(sd, pts, volumes, lookup=K.lookup_volume, second=True):
    """SDFNetwork.gradient (sdf_network.py:131-154): -> (d sdf/d x, d(sum d sdf/dx)/dx)."""
    with torch.enable_grad():
        x = pts.detach().requires_grad_(True)
        y = sdf_mlp(sd, x, volumes, lookup)[:, :1]
        g = torch.autograd.grad(y, x, torch.ones_like(y), create_graph=True)[0]
        if not second:
            return g, None
        s = torch.autograd.grad(g, x, torch.ones_like(g), create_graph=True)[0]
    return g, s


def _lin(sd, name, x):
    return x @ sd[name + ".weight"].t() + sd[name + ".bias"]


def blend_mlp(sd, rgb_feat, ray_diff, mask, prefix="color_network."):
    """BlendingNetwork.forward (blending_network.py:69-118): -> (N,3)."""
    p = prefix
    mask = mask[:, :, None].to(rgb_feat.dtype)
    s = rgb_feat.shape[1]
    dfe = F.elu(_lin(sd, p + "ray_dir_fc.2", F.elu(_lin(sd, p + "ray_dir_fc.0", ray_diff))))
    rgb_in = rgb_feat[..., :3]
    x = rgb_feat + dfe
    e = torch.exp(torch.abs(sd[p + "s"]) * (ray_diff[..., 3:4] - 1))
    w = (e - e.min(dim=1, keepdim=True)[0]) * mask
    w = w / (w.sum(dim=1, keepdim=True) + 1e-8)
    mean = (x * w).sum(1, keepdim=True)
    var = (w * (x - mean) ** 2).sum(1, keepdim=True)
    h = torch.cat([mean.expand(-1, s, -1), var.expand(-1, s, -1), x], -1)
    h = F.elu(_lin(sd, p + "base_fc.2", F.elu(_lin(sd, p + "base_fc.0", h))))
    hv = F.elu(_lin(sd, p + "vis_fc.2", F.elu(_lin(sd, p + "vis_fc.0", h * w))))
    res, vis = hv[..., :-1], hv[..., -1:]
    vis = torch.sigmoid(vis) * mask
    h = h + res
    vis = torch.sigmoid(_lin(sd, p + "vis_fc2.2", F.elu(_lin(sd, p + "vis_fc2.0", h * vis)))) * mask
    h = torch.cat([h, vis, ray_diff], -1)
    h = _lin(sd, p + "rgb_fc.4", F.elu(_lin(sd, p + "rgb_fc.2", F.elu(_lin(sd, p + "rgb_fc.0", h)))))
    h = h.masked_fill(mask == 0, -1e9)
    return (rgb_in * torch.softmax(h, dim=1)).sum(1)


def _valid_or_first10(masks, pts):
    ok = K.point_valid(masks, pts)
    if ok.sum() < 1:
        ok[:10] = True                                                  # (Q7)
    return ok


def _masked_sdf(sd, pts, volumes, masks, lookup):
    ok = _valid_or_first10(masks, pts)
    sdf = torch.full((pts.shape[0], 1), 100.0)
    sdf[ok] = sdf_mlp(sd, pts[ok], volumes, lookup)[:, :1]
    return sdf


def sample_rays(sd, rays_o, rays_d, near, far, volumes, masks, t_rand, n_samples=64, n_importance=64, steps=4,
                lookup=K.lookup_volume):
    """Coarse samples + hierarchical up-sampling (implicit_surface.py:351-393): -> z (B, n_samples+n_importance)."""
    b = rays_o.shape[0]
    z = near + (far - near) * torch.linspace(0.0, 1.0, n_samples)[None]
    z = z.expand(b, n_samples) + (t_rand - 0.5) * 2.0 / n_samples
    with torch.no_grad():
        pts = (rays_o[:, None] + rays_d[:, None] * z[..., None]).reshape(-1, 3)
        sdf = _masked_sdf(sd, pts, volumes, masks, lookup).reshape(b, n_samples)
        for i in range(steps):
            z_new = K.up_sample(rays_o, rays_d, z, sdf, n_importance // steps, masks, 64 * 2 ** i)
            if i + 1 == steps:
                z, _ = K.merge_samples(z, z_new)
            else:
                p_new = (rays_o[:, None] + rays_d[:, None] * z_new[..., None]).reshape(-1, 3)
                sdf_new = _masked_sdf(sd, p_new, volumes, masks, lookup).reshape(b, -1)
                z, sdf = K.merge_samples(z, z_new, sdf, sdf_new)
    return z


def render(sd, rays_o, rays_d, near, far, volumes, masks, imgs, features, match_features, intrs, c2ws,
           cos_anneal, step, t_rand, pts_random, n_samples=64, n_importance=64, steps=4, truncated=False, z=None):
    """ImplicitSurface.render (implicit_surface.py:351-405) -> the 18-key dict of render_core (:330-349)."""
    lookup = K.lookup_volume_truncated if truncated else K.lookup_volume
    b = rays_o.shape[0]
    n = n_samples + n_importance
    if z is None:  # tests may pin the hierarchical samples to isolate the (ill-conditioned) inverse-CDF step
        z = sample_rays(sd, rays_o, rays_d, near, far, volumes, masks, t_rand, n_samples, n_importance, steps, lookup)
    sample_dist = 2.0 / n_samples
    dists = torch.cat([z[:, 1:] - z[:, :-1], torch.full((b, 1), sample_dist)], -1)
    mid_z = z + dists * 0.5
    pts = (rays_o[:, None] + rays_d[:, None] * mid_z[..., None]).reshape(-1, 3)
    ok = _valid_or_first10(masks, pts)
    voxel_mask = (K.lookup_mask_nearest(masks, pts) > 0).any(-1).to(torch.float32).reshape(b, n)

    nn_out = sdf_mlp(sd, pts[ok], volumes, lookup)
    sdf = torch.full((b * n, 1), 100.0)
    sdf[ok] = nn_out[:, :1]
    grad_v, smooth_v = sdf_gradient(sd, pts[ok], volumes, lookup)
    gradients = torch.zeros(b * n, 3)
    smooth = torch.zeros(b * n, 3)
    gradients[ok] = grad_v
    smooth[ok] = smooth_v
    fv, rd, mv = K.lookup_feature(pts[ok], imgs, intrs, c2ws, features)
    color = torch.zeros(b * n, 3)
    color[ok] = blend_mlp(sd, fv, rd, mv)
    src_vis = torch.zeros(b * n, imgs.shape[0] - 1, dtype=torch.bool)
    src_vis[ok] = mv
    inv_s = torch.exp(sd["deviation_network.variance"] * 10.0).clamp(1e-6, 1e6)

    out = K.composite(rays_o, rays_d, z, sample_dist, sdf.reshape(b, n), gradients.reshape(b, n, 3),
                      smooth.reshape(b, n, 3), color.reshape(b, n, 3), voxel_mask, src_vis.reshape(b, n, -1),
                      inv_s, cos_anneal, c2ws[0])
    sdf_random = sdf_mlp(sd, pts_random, volumes, lookup)[:, :1]
    pts0 = out.pop("pts_sdf0")
    out.pop("mid_z")
    g0, _ = sdf_gradient(sd, pts0.reshape(-1, 3), volumes, lookup)
    g0 = g0.reshape(b, 1, 3)
    g0n = torch.linalg.norm(g0, dim=-1, keepdim=True)
    g0 = g0 / torch.where(g0n <= 0, torch.full_like(g0n, 1e-8), g0n)
    g0 = (g0 @ c2ws[0, :3, :3]).detach()                              # R^T n, as a row vector
    src = features if (step is None or step < 5) else match_features
    h, w = src[0].shape[-2:]
    warp = torch.cat([src[0].detach(), K.upsample_bilinear_half_pixel(src[1].detach(), h, w),
                      K.upsample_bilinear_half_pixel(src[2].detach(), h, w)], 1)
    ref_val, src_val = K.patch_warp(pts0, g0, warp, intrs, c2ws)
    out.update({
        "ref_gray_val": ref_val, "sampled_gray_val": src_val, "tv_reg": K.tv_regularization(volumes, masks),
        "sparse_sdf": torch.cat([sdf_random, sdf]), "gradients": gradients.reshape(b, n, 3),
        "s_val": (1.0 / inv_s).reshape(1, 1).expand(b * n, 1),
    })
    return out


def sdf_grid(sd, volumes, bound_min, bound_max, resolution, chunk=65536):
    """The lattice handed to marching cubes: u = -sdf (implicit_surface.py:407-421)."""
    pts = K.lattice_points(bound_min, bound_max, resolution)
    with torch.no_grad():
        vals = torch.cat([sdf_mlp(sd, p, volumes)[:, :1] for p in pts.split(chunk)])
    return (-vals).reshape(resolution, resolution, resolution)
