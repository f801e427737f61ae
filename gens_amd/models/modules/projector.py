"""Host-side mirror of the reference's models/modules/projector.py for the functions on the hot path
(lookup_volume :217-245, lookup_feature :294-349, surface_patch_warp :353-419, patch_homography :422-437).

Same names, argument meaning and return shapes; the arithmetic runs in libgens_hip.so (see gens_amd/ops/).
There is no CPU path: tensors must live on the MI355X.
"""
import torch

from ... import ops


def lookup_volume(pts, volume, sample_mode="grad"):
    """pts (n_pts,3) in [-1,1]; volume: one (1,C,X,Y,Z) tensor, a list of them, or a packed ops.VolumeSet.

    "grad": trilinear, align_corners=True, zeros padding, twice differentiable -> (n_pts, sum C)      (:229,238)
    "nearest": align_corners=False nearest read -> (n_pts, L) float                                     (:231,240)
    """
    pts = pts.reshape(-1, 3)
    vols = volume if isinstance(volume, (list, tuple, ops.VolumeSet, torch.nn.ParameterList)) else [volume]
    if isinstance(vols, torch.nn.ParameterList):
        vols = list(vols)
    if sample_mode == "grad":
        return ops.lookup_volume(pts, vols)
    if sample_mode == "nearest":
        return ops.lookup_mask(pts, vols, return_values=True)[1]
    raise ValueError(f"unsupported sample_mode {sample_mode!r} (the reference uses 'grad' and 'nearest')")


def lookup_feature(pts, imgs, intrs, c2ws, features, views=None):
    """-> (feat_views (N,S,3+4L), ray_diff (N,S,4), mask (N,S) bool); `views` caches the per-scene texel copies."""
    if not isinstance(features, (list, tuple)):
        features = [features]
    if views is None:
        views = ops.SceneViews(imgs, intrs, c2ws, features)
    return ops.lookup_feature(pts, views)


def patch_homography(H, uv):
    """H (B,S,3,3), uv (B,P,2) -> (S, B*P, 2) pixel coordinates in the source views (:422-437)."""
    ones = torch.ones_like(uv[..., :1])
    q = torch.einsum("bsik,bpk->sbpi", H, torch.cat([uv, ones], -1))
    q = q.reshape(H.shape[1], -1, 3)
    return q[..., :2] / (q[..., 2:] + 1e-8)


def surface_patch_warp(pts_sdf0, gradients_sdf0, images, intrinsics, poses, patch_size=11):
    """Plane-induced homography warp of a patch around each surface point (:353-419).

    pts_sdf0 (B,1,3) world points, gradients_sdf0 (B,1,3) unit normals in the reference-camera frame,
    images: (nv,C,H,W) tensor or a (texels (nv,H,W,C_pad), C) pair from ops.build_warp_features.
    -> ref_gray_val (1,B,P*P,C), sampled_gray_val (S,B,P*P,C); differentiable w.r.t. pts_sdf0.
    """
    if isinstance(images, tuple):
        tex, c = images
    else:
        tex, c = ops.pack_nchw(images.detach()), images.shape[1]
    nv, h, w, _ = tex.shape
    b = pts_sdf0.shape[0]
    r_ref, c_ref = poses[0, :3, :3], poses[0, :3, 3]
    k_ref = intrinsics[0, :3, :3]
    k_ref_inv = ops.SceneCams.of(intrinsics, poses).kinv_ref
    p, nrm = pts_sdf0[:, 0], gradients_sdf0[:, 0]
    x_cam = p @ r_ref + (-(r_ref.t() @ c_ref))[None]
    proj = x_cam @ k_ref.t()
    disp = (nrm * x_cam).sum(-1)
    r_src_t = poses[1:, :3, :3].transpose(1, 2)
    rel = r_src_t @ r_ref
    tvec = (r_src_t @ (c_ref[None] - poses[1:, :3, 3])[..., None])[..., 0]
    hom = rel[None] + (tvec[None, :, :, None] * nrm[:, None, None, :]) / (disp[:, None, None, None] + 1e-10)
    hom = intrinsics[None, 1:, :3, :3] @ hom @ k_ref_inv[None, None]

    u0 = proj[:, 0] / (proj[:, 2] + 1e-8)
    v0 = proj[:, 1] / (proj[:, 2] + 1e-8)
    half = patch_size // 2
    offs = torch.arange(-half, half + 1, device=p.device, dtype=p.dtype)
    oy, ox = torch.meshgrid(offs, offs, indexing="ij")
    uv = torch.stack([u0[:, None] + ox.reshape(-1)[None], v0[:, None] + oy.reshape(-1)[None]], -1)   # (B,P,2)
    n_px = uv.shape[1]

    grid = patch_homography(hom, uv)                                                                     # (S,B*P,2)
    # the reference normalises with (w-1)/2 and reads with align_corners=True: keep the round trip for its rounding
    gx = ((2 * grid[..., 0] / (w - 1) - 1.0) + 1) / 2 * (w - 1)
    gy = ((2 * grid[..., 1] / (h - 1) - 1.0) + 1) / 2 * (h - 1)
    xy_src = torch.stack([gx, gy], -1)
    sampled = torch.stack([ops.patch_sample(tex[v + 1], xy_src[v], c) for v in range(nv - 1)], 0)
    sampled = sampled.reshape(nv - 1, b, n_px, c)
    uvd = uv.detach()
    rx = ((2 * uvd[..., 0] / (w - 1) - 1.0) + 1) / 2 * (w - 1)
    ry = ((2 * uvd[..., 1] / (h - 1) - 1.0) + 1) / 2 * (h - 1)
    ref = ops.patch_sample(tex[0], torch.stack([rx, ry], -1).reshape(-1, 2), c).reshape(1, b, n_px, c)
    return ref, sampled
