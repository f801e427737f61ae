"""Seeded synthetic DTU-like scenes: cameras, images, feature pyramid, rays.

The reference has no data offline, so every test / benchmark input is generated here.
Camera conventions follow what the reference datasets hand to the model
(/root/reference/datasets/dtu.py:330-341, 390-403): scene normalised to the unit sphere,
view 0 is the reference view, ``c2ws`` are camera-to-world, ``intrs`` are 4x4 with the
3x3 pinhole matrix in the top-left block, rays leave the reference camera centre through
pixel centres at integer coordinates.
"""
import math

import numpy as np
import torch

# DTU 1600x1200 calibration scaled to 640x480.
DTU_FX, DTU_FY, DTU_CX, DTU_CY = 1156.93, 1153.27, 329.28, 247.63


def _rot_y(deg):
    a = math.radians(deg)
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], dtype=np.float64)


def make_cameras(nv, h=480, w=640, dist=2.2):
    """Reference camera at (0,0,-dist) looking at the origin, sources rotated about y.

    Returns intrs (nv,4,4), c2ws (nv,4,4), near (1,1), far (1,1) as float32 tensors.
    Intrinsics are those of a 480x640 image rescaled to (h, w).
    """
    angles = [0.0 if k == 0 else (10.0 if k % 2 else -10.0) * ((k + 1) // 2) for k in range(nv)]     # 0, 10, -10, 20, -20, ...: any view count
    sx, sy = w / 640.0, h / 480.0
    intr = np.eye(4, dtype=np.float64)
    intr[0, 0], intr[1, 1] = DTU_FX * sx, DTU_FY * sy
    intr[0, 2], intr[1, 2] = DTU_CX * sx, DTU_CY * sy
    intrs, c2ws = [], []
    for a in angles:
        r = _rot_y(a)
        c2w = np.eye(4, dtype=np.float64)
        c2w[:3, :3] = r
        c2w[:3, 3] = r @ np.array([0.0, 0.0, -dist])
        intrs.append(intr.copy())
        c2ws.append(c2w)
    near = torch.tensor([[0.95 * (dist - 1.0)]], dtype=torch.float32)
    far = torch.tensor([[1.05 * (dist + 1.0)]], dtype=torch.float32)
    return (torch.from_numpy(np.stack(intrs)).float(), torch.from_numpy(np.stack(c2ws)).float(), near, far)


def make_rays(intrs, c2ws, h, w, step=1, pixels=None):
    """Rays of view 0 (dtu.py:390-403). ``pixels`` (n,2) of (x,y) overrides the regular lattice."""
    if pixels is None:
        tx = torch.linspace(0, w - 1, w // step)
        ty = torch.linspace(0, h - 1, h // step)
        py, px = torch.meshgrid(ty, tx, indexing="ij")
        px, py = px.reshape(-1), py.reshape(-1)
    else:
        px, py = pixels[:, 0].float(), pixels[:, 1].float()
    p = torch.stack([px, py, torch.ones_like(py)], dim=-1)
    p = (torch.inverse(intrs[0, :3, :3])[None] @ p[:, :, None])[:, :, 0]
    d = p / torch.linalg.norm(p, dim=-1, keepdim=True)
    rays_d = (c2ws[0, :3, :3][None] @ d[:, :, None])[:, :, 0]
    rays_o = c2ws[0, :3, 3][None].expand_as(rays_d).contiguous()
    return rays_o.contiguous(), rays_d.contiguous()


def make_scene(nv=5, h=480, w=640, n_levels=5, channels=4, seed=0, dist=2.2):
    """Cameras -> imgs -> features, drawn in that order from one seeded generator."""
    g = torch.Generator().manual_seed(seed)
    intrs, c2ws, near, far = make_cameras(nv, h, w, dist)
    imgs = torch.rand(nv, 3, h, w, generator=g)
    feats = [torch.randn(nv, channels, h >> i, w >> i, generator=g) for i in range(n_levels)]
    return {"intrs": intrs, "c2ws": c2ws, "near": near, "far": far, "imgs": imgs, "features": feats, "hw": (h, w)}


def make_volumes(dims, channels=4, scale=0.1, seed=1):
    """Stand-in for the regularised volumes (reg_network is out of scope)."""
    g = torch.Generator().manual_seed(seed)
    return [scale * torch.randn(1, channels, d, d, d, generator=g) for d in dims]
