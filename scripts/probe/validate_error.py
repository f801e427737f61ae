#!/usr/bin/env python3
"""L1 error of validate() (fused inference path, chunk 512) against the reference's own validate (golden g15), per arithmetic."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from tests.test_hip_render import build_surface, scene_inputs
g = {k: torch.as_tensor(v) if v.dtype.kind != "U" else v for k, v in np.load("tests/golden/g15_validate.npz").items()}
g["step"] = torch.tensor(-1.0)
c = lambda t: t.cuda()
for prec in ("f32", "f16x2"):
    surf = build_surface(g); feats, vols, masks, match, _ = scene_inputs(g)
    surf.val_chunk = 512; surf.sdf_precision = prec
    torch.manual_seed(int(g["rng_seed"]))
    out = surf.validate(c(g["rays_o"]), c(g["rays_d"]), c(g["near"]), c(g["far"]), vols, masks, c(g["imgs"]), feats, match, c(g["intrs"]), c(g["c2ws"]),
                        None, None, torch.tensor([24, 32]).int(), extract_geometry=False)
    print(prec, {k: float((torch.as_tensor(out[k]) - g["out." + k]).abs().mean()) for k in ("color_fine", "render_depth", "sdf_depth", "normal_img")})
