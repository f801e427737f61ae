#!/usr/bin/env python3
"""Generate golden vectors by running the reference's own Python on CPU (this container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Imports /root/reference with the shims of SURVEY.md §8c (nothing from the reference is copied
into the repo; only tensors -- inputs and the outputs the reference computed -- are stored):

* stub modules ``mcubes``, ``torchvision`` and ``models.modules.grid_sample_cuda.cuda_gridsample``
  are inserted in ``sys.modules`` before the import (the real ones need PyMCubes / torchvision /
  nvcc, all absent);
* ``cug.grid_sample_3d`` is the reference's OWN autograd Function pair (cuda_gridsample.py:71-123),
  executed from its source; only its CUDA entry point ``grad2_3d`` is replaced, by autograd over
  ``_sampler3d_zeros`` below (a zeros-padding trilinear sampler written with differentiable torch
  ops), see ``_reference_function_pair``.  The stand-in is checked here against ``F.grid_sample``
  (value + 1st order, every point) and against the reference's own pure-torch
  ``projector.grid_sample_3d`` (projector.py:62-214; 2nd order, in-cube points) before any golden
  is written.  Third derivatives through the sampler are dropped, as on the reference's GPU path;
* ``torch.Tensor.cuda`` is made the identity (implicit_surface.py:270 hard-codes ``.cuda()``);
* a dict subclass stands in for pyhocon's ConfigTree.

Outputs: tests/golden/*.npz (a few MB in total).
"""
import os
import sys
import types

sys.dont_write_bytecode = True
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)
from gens_amd import synthetic  # noqa: E402

torch.set_num_threads(8)


# --------------------------------------------------------------------------------------
# shims
# --------------------------------------------------------------------------------------
def _sampler3d_zeros(input, grid, padding_mode="zeros", align_corners=True):
    """Trilinear, zeros padding, align_corners=True; differentiable to any order in torch."""
    assert padding_mode == "zeros" and align_corners
    n, c, d, h, w = input.shape
    assert n == 1
    g = grid.reshape(-1, 3)
    size = torch.tensor([w, h, d], dtype=g.dtype)
    pos = (g + 1) * 0.5 * (size - 1)
    base = torch.floor(pos.detach())
    frac = pos - base
    base = base.long()
    flat = input.reshape(c, -1)
    out = 0
    for dz in (0, 1):
        for dy in (0, 1):
            for dx in (0, 1):
                ix, iy, iz = base[:, 0] + dx, base[:, 1] + dy, base[:, 2] + dz
                wx = frac[:, 0] if dx else 1 - frac[:, 0]
                wy = frac[:, 1] if dy else 1 - frac[:, 1]
                wz = frac[:, 2] if dz else 1 - frac[:, 2]
                ok = (ix >= 0) & (ix < w) & (iy >= 0) & (iy < h) & (iz >= 0) & (iz < d)
                lin = (iz.clamp(0, d - 1) * h + iy.clamp(0, h - 1)) * w + ix.clamp(0, w - 1)
                val = flat[:, lin] * ok.to(input.dtype)[None]
                out = out + val * (wx * wy * wz)[None]
    return out.reshape(1, c, 1, 1, -1)


def _grad2_3d_cpu(gg_input, gg_grid, g_out, input, grid, padding_mode, align_corners):
    """CPU stand-in for the reference's CUDA entry point `gridsample_grad2.grad2_3d` (gridsample_cuda.cpp:42-56, kernel
    gridsample_cuda.cu:212-533): the derivative of <gI, ggI> + <gG, ggG> -- (gI, gG) = first-order backward of the trilinear read --
    with respect to (gO, input, grid), here by autograd over `_sampler3d_zeros`.  Plain tensors come back, exactly as from the CUDA op."""
    if padding_mode or not align_corners:
        raise RuntimeError("only zeros padding / align_corners=True is exercised by the reference (projector.py:229,238)")
    with torch.enable_grad():
        o = g_out.detach().clone().requires_grad_(True)
        v = input.detach().clone().requires_grad_(True)
        x = grid.detach().clone().requires_grad_(True)
        y = _sampler3d_zeros(v, x).reshape(o.shape)
        g_v, g_x = torch.autograd.grad(y, [v, x], o, create_graph=True)
        phi = (g_v * gg_input.detach()).sum() + (g_x * gg_grid.detach()).sum()
        outs = torch.autograd.grad(phi, [o, v, x], allow_unused=True)
    return [t.detach() if t is not None else torch.zeros_like(z) for t, z in zip(outs, (o, v, x))]


def _reference_function_pair():
    """The reference's OWN autograd pair `_GridSample3dForward` / `_GridSample3dBackward` (cuda_gridsample.py:71-123), executed from its
    source file: forward = F.grid_sample, first backward = aten::grid_sampler_3d_backward, second backward = `grad2_3d`, whose outputs are
    plain tensors -- so anything differentiated a THIRD time (the training step's loss.backward() through `smooth`, sdf_network.py:146)
    sees them as constants.  That truncation is the reference's behaviour on the GPU and the goldens must carry it: a fully
    differentiable stand-in sampler does not (round 1's goldens used one; the L = 5 goldens of round 2 exposed the difference in
    d loss / d weight_v of lin1..lin6).  Two things cannot run here and are replaced: the JIT build of the CUDA extension at import
    (`cpp_extension.load` returns the CPU stand-in above) and the `.is_cuda` asserts of the second backward (the module is compiled
    with optimize=1, which strips `assert`)."""
    from torch.utils import cpp_extension
    path = os.path.join(REF, "models/modules/grid_sample_cuda/cuda_gridsample.py")
    with open(path) as f:
        code = compile(f.read(), path, "exec", optimize=1)
    mod = types.ModuleType("models.modules.grid_sample_cuda.cuda_gridsample")
    mod.__file__ = path
    real_load = cpp_extension.load
    cpp_extension.load = lambda *a, **k: types.SimpleNamespace(grad2_3d=_grad2_3d_cpu, grad2_2d=None)
    try:
        exec(code, mod.__dict__)
    finally:
        cpp_extension.load = real_load
    return mod


def _install_shims():
    for name in ("mcubes", "torchvision", "torchvision.models"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["torchvision"].models = sys.modules["torchvision.models"]
    sys.modules["mcubes"].marching_cubes = lambda u, t: (np.zeros((0, 3)), np.zeros((0, 3), dtype=np.int64))
    cug = _reference_function_pair()
    sys.modules["models.modules.grid_sample_cuda.cuda_gridsample"] = cug
    pkg = types.ModuleType("models.modules.grid_sample_cuda")
    pkg.__path__ = []
    pkg.cuda_gridsample = cug
    sys.modules["models.modules.grid_sample_cuda"] = pkg
    torch.Tensor.cuda = lambda self, *a, **k: self
    sys.path.insert(0, REF)
    os.chdir("/tmp")


class Conf(dict):
    """Duck-typed pyhocon ConfigTree: dotted keys + get_int/get_float/get_list/get_bool."""

    def _walk(self, key):
        node = self
        for part in key.split("."):
            node = dict.__getitem__(node, part)
        return node

    def __getitem__(self, key):
        v = self._walk(key)
        return Conf(v) if isinstance(v, dict) and not isinstance(v, Conf) else v

    def get(self, key, default=None):
        try:
            return self[key]
        except KeyError:
            return default

    get_int = get_float = get_list = get_bool = get


def surf_conf(n_levels_vol, n_levels_feat):
    return Conf({
        "sdf_network": dict(d_out=129, d_in=3, d_hidden=128, n_layers=6, skip_in=[3], multires=4, bias=0.5,
                            scale=1.0, geometric_init=True, weight_norm=True, feat_channels=4 * n_levels_vol),
        "color_network": dict(d_feature=4 * n_levels_feat),
        "variance_network": dict(init_val=0.3),
        "render": dict(n_samples=64, n_importance=64, up_sample_steps=4, perturb=1.0),
    })


def npz(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {name}.npz  {os.path.getsize(path) / 1e6:.2f} MB")


# --------------------------------------------------------------------------------------
# goldens
# --------------------------------------------------------------------------------------
def g1_volume(Volume):
    # (a) BASELINE config 1: 3 views, coarsest (16^3) volume only, level-4 map, intrinsics * 2^-4 (Q2)
    sc = synthetic.make_scene(nv=3, h=480, w=640, n_levels=5, seed=11)
    intr = sc["intrs"].clone()
    intr[:, :2] *= 0.5 ** 4
    feat = sc["features"][4].clone().requires_grad_(True)
    vol = Volume(Conf({"volume_dims": [16]}))
    v, m = vol.agg_mean_var([feat], intr, sc["c2ws"])
    g = torch.Generator().manual_seed(5)
    cot = torch.randn(v[0].shape, generator=g)
    (v[0] * cot).sum().backward()
    npz("g1a_volume_c1", feat=feat, intrs=intr, c2ws=sc["c2ws"], volume=v[0], mask=m[0], cot=cot, gfeat=feat.grad)

    # (b) 5 views, three scales, 60x80 pyramid
    sc = synthetic.make_scene(nv=5, h=60, w=80, n_levels=3, seed=12)
    feats = [f.clone().requires_grad_(True) for f in sc["features"]]
    dims = [24, 12, 6]
    vol = Volume(Conf({"volume_dims": dims}))
    v, m = vol.agg_mean_var(feats, sc["intrs"], sc["c2ws"])
    cots = [torch.randn(x.shape, generator=g) for x in v]
    sum((a * b).sum() for a, b in zip(v, cots)).backward()
    d = dict(intrs=sc["intrs"], c2ws=sc["c2ws"], dims=np.array(dims))
    for i in range(3):
        d.update({f"feat{i}": feats[i], f"volume{i}": v[i], f"mask{i}": m[i], f"cot{i}": cots[i], f"gfeat{i}": feats[i].grad})
    npz("g1b_volume_ms", **d)


def g1c_volume_tiles(Volume):
    """K1 backward over several image tiles of the image-tile kernel (64 x 30 texels): 4 views 96 x 160 (3 x 4 tiles), one 32^3 volume."""
    sc = synthetic.make_scene(nv=4, h=96, w=160, n_levels=1, seed=13)
    feat = sc["features"][0].clone().requires_grad_(True)
    vol = Volume(Conf({"volume_dims": [32]}))
    v, m = vol.agg_mean_var([feat], sc["intrs"], sc["c2ws"])
    g = torch.Generator().manual_seed(6)
    cot = torch.randn(v[0].shape, generator=g)
    (v[0] * cot).sum().backward()
    npz("g1c_volume_tiles", feat=feat, intrs=sc["intrs"], c2ws=sc["c2ws"], mask=m[0], cot=cot, gfeat=feat.grad)


def g2_lookup(projector):
    g = torch.Generator().manual_seed(21)
    dims = [12, 8, 5]
    vols = [torch.randn(1, 4, d, d, d, generator=g).requires_grad_(True) for d in dims]
    pts = (torch.rand(300, 3, generator=g) * 2.6 - 1.3)
    pts[:8] = torch.tensor([[1, 1, 1], [-1, -1, -1], [1, -1, 0.3], [0, 0, 0], [1.0001, 0, 0], [-1.0001, 0.5, 0.5],
                            [0.999, 0.999, -0.999], [0.2, -1, 1]], dtype=torch.float32)
    pts.requires_grad_(True)
    incube = (pts.detach().abs() < 0.999).all(-1)

    # value + first order from the op the reference's forward/backward really calls (F.grid_sample /
    # aten::grid_sampler_3d_backward; cuda_gridsample.py:79,97)
    x = pts.unsqueeze(0).unsqueeze(0).unsqueeze(0).flip(dims=[-1])
    feats = torch.cat([F.grid_sample(v, x, padding_mode="zeros", align_corners=True).reshape(-1, 300).permute(1, 0)
                       for v in vols], -1)
    gO = torch.randn(feats.shape, generator=g)
    grads = torch.autograd.grad(feats, vols + [pts], gO)
    gV, gP = grads[:3], grads[3]

    # the stand-in sampler behind grad2_3d: value and first order must be ATen's at every point (in and out of the cube)
    feats_z = torch.cat([_sampler3d_zeros(v, x).reshape(-1, 300).permute(1, 0) for v in vols], -1)
    assert torch.allclose(feats_z, feats, atol=1e-6), "stand-in sampler forward != F.grid_sample"
    gz = torch.autograd.grad(feats_z, vols + [pts], gO)
    assert torch.allclose(gz[3], gP, atol=1e-5), "stand-in sampler d/dpts != aten backward"
    assert all(torch.allclose(a, b, atol=1e-5) for a, b in zip(gz[:3], gV)), "stand-in sampler d/dvolume != aten backward"
    # same through lookup_volume on the reference's Function pair (the path the model takes) -> must agree
    feats_s = projector.lookup_volume(pts, vols)
    assert torch.equal(feats_s, feats)
    gp_s = torch.autograd.grad(feats_s, pts, gO, create_graph=True)[0]
    assert torch.allclose(gp_s, gP, atol=1e-6)

    # second order: cotangent ggG on gp  ->  (ggO, gV', gp')   [what grad2_3d returns, gridsample_cuda.cpp:42-56]
    ggG = torch.randn(300, 3, generator=g)
    gO_leaf = gO.clone().requires_grad_(True)
    gp_l = torch.autograd.grad(projector.lookup_volume(pts, vols), pts, gO_leaf, create_graph=True)[0]
    outs = torch.autograd.grad(gp_l, [gO_leaf] + vols + [pts], ggG, allow_unused=True)
    ggO, gV2, gP2 = outs[0], outs[1:4], outs[4]

    # cross-check the in-cube part against the reference's own pure-torch sampler (projector.py:62-214)
    def ref_lookup(p, vs):
        xx = p.unsqueeze(0).unsqueeze(0).unsqueeze(0).flip(dims=[-1])
        return torch.cat([projector.grid_sample_3d(v, xx).reshape(-1, p.shape[0]).permute(1, 0) for v in vs], -1)
    pin = pts.detach()[incube].clone().requires_grad_(True)
    gO_in = gO[incube].clone().requires_grad_(True)
    gp_r = torch.autograd.grad(ref_lookup(pin, vols), pin, gO_in, create_graph=True)[0]
    outs_r = torch.autograd.grad(gp_r, [gO_in, pin], ggG[incube])
    assert torch.allclose(outs_r[0], ggO[incube], atol=1e-4), "2nd order ggO differs from reference pure-torch sampler"
    assert torch.allclose(outs_r[1], gP2[incube], atol=1e-4), "2nd order gp' differs from reference pure-torch sampler"

    d = dict(pts=pts, feats=feats, gO=gO, gP=gP, ggG=ggG, ggO=ggO, gP2=gP2, incube=incube, dims=np.array(dims))
    for i in range(3):
        d.update({f"vol{i}": vols[i], f"gV{i}": gV[i], f"gV2_{i}": gV2[i]})
    npz("g2_lookup", **d)


def g3_nearest(projector):
    g = torch.Generator().manual_seed(31)
    dims = [12, 8, 5]
    masks = [(torch.rand(1, 1, d, d, d, generator=g) > 0.5).float() for d in dims]
    pts = torch.rand(400, 3, generator=g) * 2.4 - 1.2
    # half-integer ties of the align_corners=False index ((p+1)*D-1)/2 = k+.5  (Q6)
    k = 0
    for d in dims:
        for idx in range(d):
            pts[k % 200, k % 3] = (2.0 * idx + 2.0) / d - 1.0
            k += 1
    pts[200:206] = torch.tensor([[1, 1, 1], [-1, -1, -1], [1, 0, -1], [0.99999, 0, 0], [-1, 1, 0], [0, 0, 0]], dtype=torch.float32)
    val = projector.lookup_volume(pts, masks, sample_mode="nearest")
    d = dict(pts=pts, val=val, any=val.any(dim=-1), dims=np.array(dims))
    for i in range(3):
        d[f"mask{i}"] = masks[i]
    npz("g3_nearest", **d)


def g4_feature(projector):
    sc = synthetic.make_scene(nv=4, h=48, w=64, n_levels=5, seed=41)
    g = torch.Generator().manual_seed(42)
    pts = torch.rand(256, 3, generator=g) * 2 - 1
    pts[:4] = torch.tensor([[0, 0, -3.0], [0, 0, -2.2], [2.5, 0, -2.0], [0.9, 0.9, 0.9]])
    feats = [f.clone().requires_grad_(True) for f in sc["features"]]
    imgs = sc["imgs"].clone().requires_grad_(True)
    fv, rd, mk = projector.lookup_feature(pts, imgs, sc["intrs"], sc["c2ws"], feats)
    cot = torch.randn(fv.shape, generator=g)
    fv_f = torch.nan_to_num(fv, nan=0.0, posinf=0.0, neginf=0.0)
    grads = torch.autograd.grad((fv_f * cot).sum(), feats + [imgs])
    d = dict(pts=pts, imgs=imgs, intrs=sc["intrs"], c2ws=sc["c2ws"], feat_views=fv, ray_diff=rd, mask=mk, cot=cot, gimgs=grads[5])
    for i in range(5):
        d.update({f"feat{i}": feats[i], f"gfeat{i}": grads[i]})
    npz("g4_feature", **d)


def g5_upsample(isurf_mod, projector):
    g = torch.Generator().manual_seed(51)
    surf = isurf_mod.ImplicitSurface(surf_conf(3, 5))
    sc = synthetic.make_scene(nv=3, h=48, w=64, n_levels=5, seed=52)
    rays_o, rays_d = synthetic.make_rays(sc["intrs"], sc["c2ws"], 48, 64, pixels=torch.tensor([[3, 5], [20, 20], [32, 24], [40, 30], [60, 44], [31, 23], [10, 40]]))
    b = rays_o.shape[0]
    dims = [16, 8, 4]
    masks = [(torch.rand(1, 1, d, d, d, generator=g) > 0.3).float() for d in dims]
    d = dict(rays_o=rays_o, rays_d=rays_d, dims=np.array(dims))
    for i in range(3):
        d[f"mask{i}"] = masks[i]
    for r, n in enumerate([64, 80, 96, 112]):
        z = torch.sort(torch.rand(b, n, generator=g) * 2.2 + 1.1, dim=-1)[0]
        pts = rays_o[:, None] + rays_d[:, None] * z[..., None]
        sdf = torch.linalg.norm(pts, dim=-1) - 0.6 + 0.05 * torch.randn(b, n, generator=g)
        sdf[0, 5:9] = 100.0  # masked-out samples carry the constant 100 (Q8)
        inv_s = 64 * 2 ** r
        zs = surf.up_sample(rays_o, rays_d, z, sdf, 16, masks, inv_s)
        z2, _ = surf.cat_z_vals(rays_o, rays_d, z, zs, sdf, None, masks, last=True)
        d.update({f"z{r}": z, f"sdf{r}": sdf, f"znew{r}": zs, f"zcat{r}": z2})
    # sample_pdf alone (det=True), incl. a flat pdf and an all-zero weight row
    bins = torch.sort(torch.rand(5, 33, generator=g), dim=-1)[0]
    w = torch.rand(5, 32, generator=g)
    w[1] = 0.0
    w[2] = 1.0
    w[3, :30] = 0.0
    d.update(pdf_bins=bins, pdf_w=w, pdf_out=isurf_mod.sample_pdf(bins, w, 16, det=True))
    npz("g5_upsample", **d)


def g7_patchwarp(projector):
    sc = synthetic.make_scene(nv=3, h=48, w=64, n_levels=1, channels=12, seed=71)
    g = torch.Generator().manual_seed(72)
    imgs = sc["features"][0]
    pts = (torch.rand(10, 1, 3, generator=g) - 0.5) * 0.8
    pts[0, 0] = sc["c2ws"][0, :3, 3]  # "no crossing" rays sample at the camera centre (implicit_surface.py:301-305)
    pts.requires_grad_(True)
    nrm = torch.randn(10, 1, 3, generator=g)
    nrm = nrm / torch.linalg.norm(nrm, dim=-1, keepdim=True)
    ref, smp = projector.surface_patch_warp(pts, nrm, imgs, sc["intrs"], sc["c2ws"])
    cot = torch.randn(smp.shape, generator=g)
    gp = torch.autograd.grad((smp[:, 1:] * cot[:, 1:]).sum(), pts)[0]
    npz("g7_patchwarp", pts=pts, normals=nrm, images=imgs, intrs=sc["intrs"], c2ws=sc["c2ws"], ref_val=ref, src_val=smp, cot=cot, gpts=gp)


def g8_tv(isurf_mod):
    g = torch.Generator().manual_seed(81)
    surf = isurf_mod.ImplicitSurface(surf_conf(2, 5))
    dims = [9, 5]
    vols = [torch.randn(1, 4, d, d, d, generator=g).requires_grad_(True) for d in dims]
    masks = [(torch.rand(1, 1, d, d, d, generator=g) > 0.3).float() for d in dims]
    tv = surf.tv_regularization(vols, masks)
    gv = torch.autograd.grad(tv, vols)
    npz("g8_tv", vol0=vols[0], vol1=vols[1], mask0=masks[0], mask1=masks[1], tv=tv, gvol0=gv[0], gvol1=gv[1], dims=np.array(dims))


def _perturb(module, seed, scale):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in module.parameters():
            p.add_(scale * torch.randn(p.shape, generator=g) * (p.abs().mean() + 0.02))


def g9_render(isurf_mod, Volume, tag, seed, cos_anneal, step, n_rays, variance=0.3, nv=3, dims=(24, 16, 8)):
    """End-to-end ImplicitSurface.render (all 18 keys) + recorded intermediates.  nv = 5, five dims: the shipped level count
    (confs/gens.conf:63-67,86) with four source views."""
    torch.manual_seed(seed)
    h, w = 48, 64
    sc = synthetic.make_scene(nv=nv, h=h, w=w, n_levels=5, seed=seed)
    dims = list(dims)
    nl = len(dims)
    vols = synthetic.make_volumes(dims, seed=seed + 1)
    with torch.no_grad():
        _, masks = Volume(Conf({"volume_dims": dims})).agg_mean_var(sc["features"][:nl], sc["intrs"], sc["c2ws"])
    surf = isurf_mod.ImplicitSurface(surf_conf(nl, 5))
    _perturb(surf.sdf_network, seed + 2, 0.04)
    with torch.no_grad():
        surf.deviation_network.variance.fill_(variance)
    _perturb(surf.color_network, seed + 3, 0.05)
    g = torch.Generator().manual_seed(seed + 4)
    pix = torch.stack([torch.randint(0, w, (n_rays,), generator=g), torch.randint(0, h, (n_rays,), generator=g)], -1)
    rays_o, rays_d = synthetic.make_rays(sc["intrs"], sc["c2ws"], h, w, pixels=pix)
    match_feats = [f + 0.01 for f in sc["features"]]

    rec = {}
    orig_core = surf.render_core

    def core(rays_o_, rays_d_, z_vals, *a, **k):
        rec["z_final"] = z_vals.detach().clone()
        return orig_core(rays_o_, rays_d_, z_vals, *a, **k)
    surf.render_core = core
    orig_rand = torch.rand
    draws = []

    def rand(*a, **k):
        r = orig_rand(*a, **k)
        draws.append(r.clone())
        return r
    torch.rand = rand
    torch.manual_seed(seed + 100)
    try:
        out = surf.render(rays_o, rays_d, sc["near"], sc["far"], vols, masks, sc["imgs"], sc["features"], match_feats,
                          sc["intrs"], sc["c2ws"], cos_anneal, step)
    finally:
        torch.rand = orig_rand
    d = dict(rays_o=rays_o, rays_d=rays_d, near=sc["near"], far=sc["far"], imgs=sc["imgs"], intrs=sc["intrs"], c2ws=sc["c2ws"],
             dims=np.array(dims), cos_anneal=np.float32(cos_anneal), step=np.float32(-1 if step is None else step),
             rng_seed=np.int64(seed + 100), draw_trand=draws[0], draw_ptsrand=draws[1], z_final=rec["z_final"])
    for i in range(5):
        d[f"feat{i}"] = sc["features"][i]
    for i in range(nl):
        d[f"vol{i}"] = vols[i]
        d[f"mask{i}"] = masks[i]
    for k, v in surf.state_dict().items():
        d["sd." + k] = v
    for k, v in out.items():
        d["out." + k] = v
    npz(tag, **d)
    return surf, sc, vols, masks


def g9d_config0(isurf_mod, Volume):
    """BASELINE config[0] as written: 3 views 480 x 640, the coarsest volume only (16^3 paired with the level-4 map and intrinsics * 2^-4,
    Q2), 512 rays through ImplicitSurface.render.  The scene is synthetic.make_scene(seed) (regenerated by the test: 3 x 480 x 640 images
    are not stored); stored: seeds, rays, the K1 mask, the volume, weights, outputs (patch tensors for the first 16 rays only)."""
    seed, n_rays = 300, 512
    torch.manual_seed(seed)
    h, w, nv = 480, 640, 3
    sc = synthetic.make_scene(nv=nv, h=h, w=w, n_levels=5, seed=seed)
    intr4 = sc["intrs"].clone()
    intr4[:, :2] *= 0.5 ** 4
    with torch.no_grad():
        _, masks = Volume(Conf({"volume_dims": [16]})).agg_mean_var([sc["features"][4]], intr4, sc["c2ws"])
    vols = synthetic.make_volumes([16], seed=seed + 1)
    surf = isurf_mod.ImplicitSurface(surf_conf(1, 5))
    _perturb(surf.sdf_network, seed + 2, 0.04)
    _perturb(surf.color_network, seed + 3, 0.05)
    g = torch.Generator().manual_seed(seed + 4)
    pix = torch.stack([torch.randint(0, w, (n_rays,), generator=g), torch.randint(0, h, (n_rays,), generator=g)], -1)
    rays_o, rays_d = synthetic.make_rays(sc["intrs"], sc["c2ws"], h, w, pixels=pix)
    rec = {}
    orig_core = surf.render_core

    def core(rays_o_, rays_d_, z_vals, *a, **k):
        rec["z_final"] = z_vals.detach().clone()
        return orig_core(rays_o_, rays_d_, z_vals, *a, **k)
    surf.render_core = core
    orig_rand, draws = torch.rand, []

    def rand(*a, **k):
        r = orig_rand(*a, **k)
        draws.append(r.clone())
        return r
    torch.rand = rand
    torch.manual_seed(seed + 100)
    try:
        out = surf.render(rays_o, rays_d, sc["near"], sc["far"], vols, masks, sc["imgs"], sc["features"], sc["features"],
                          sc["intrs"], sc["c2ws"], 1.0, None)
    finally:
        torch.rand = orig_rand
    d = dict(scene_seed=np.int64(seed), pix=pix, rays_o=rays_o, rays_d=rays_d, rng_seed=np.int64(seed + 100), draw_trand=draws[0],
             draw_ptsrand=draws[1], z_final=rec["z_final"], vol0=vols[0], mask0=masks[0],
             feat_sums=np.array([float(f.double().sum()) for f in sc["features"]] + [float(sc["imgs"].double().sum())]))
    for k, v in surf.state_dict().items():
        d["sd." + k] = v
    for k, v in out.items():
        d["out." + k] = v[:, :16] if k in ("ref_gray_val", "sampled_gray_val") else v
    npz("g9d_config0", **d)


def g10_geometry(surf, vols):
    import mcubes
    grabbed = {}
    mcubes.marching_cubes = lambda u, t: (grabbed.setdefault("u", u.copy()), (np.zeros((1, 3)), np.zeros((0, 3), dtype=np.int64)))[1]
    bmin, bmax = torch.tensor([-1.0, -1.0, -1.0]), torch.tensor([1.0, 1.0, 1.0])
    surf.extract_geometry(vols, bmin, bmax, 65, 0.0)
    npz("g10_geometry", u=grabbed["u"].astype(np.float32), resolution=np.int64(65))


def g15_validate(surf, sc, vols, masks, tag="g15_validate"):
    """The reference's ImplicitSurface.validate (implicit_surface.py:429-470) on a 24 x 32 image: three 256-ray chunks, each drawing
    its jitter and its 1024 random points from the CPU generator, image assembly, normal rotation, the * 256 / * 128 + 128 scalings
    and clips (Q15); the SDF lattice handed to marching cubes (PyMCubes is absent: a stub records it)."""
    import mcubes
    grabbed = {}
    mcubes.marching_cubes = lambda u, t: (grabbed.setdefault("u", u.copy()), (np.zeros((1, 3)), np.zeros((0, 3), dtype=np.int64)))[1]
    h, w = 48, 64
    rays_o, rays_d = synthetic.make_rays(sc["intrs"], sc["c2ws"], h, w, step=2)
    bmin, bmax = torch.tensor([-1.0, -1.0, -1.0]), torch.tensor([1.0, 1.0, 1.0])
    match_feats = [f + 0.01 for f in sc["features"]]
    torch.manual_seed(1500)
    with torch.no_grad():
        out = surf.validate(rays_o, rays_d, sc["near"], sc["far"], vols, masks, sc["imgs"], sc["features"], match_feats, sc["intrs"], sc["c2ws"],
                            bmin, bmax, (h // 2, w // 2), cos_anneal_ratio=1.0, step=None, extract_geometry=True, mesh_resolution=33, threshold=0.0)
    d = dict(rays_o=rays_o, rays_d=rays_d, near=sc["near"], far=sc["far"], imgs=sc["imgs"], intrs=sc["intrs"], c2ws=sc["c2ws"],
             rng_seed=np.int64(1500), u=grabbed["u"].astype(np.float32), hw=np.array([h // 2, w // 2]))
    for i in range(5):
        d[f"feat{i}"] = sc["features"][i]
    for i in range(len(vols)):
        d[f"vol{i}"] = vols[i]
        d[f"mask{i}"] = masks[i]
    for k, v in surf.state_dict().items():
        d["sd." + k] = v
    for k in ("color_fine", "img_fine", "normal_img", "sdf_depth", "render_depth"):
        d["out." + k] = out[k]
    npz(tag, **d)


def g11_lncc():
    """compute_LNCC (models/losses/ncc.py:7-55): forward + gradients w.r.t. both patch tensors for a fixed cotangent."""
    from models.losses.ncc import compute_LNCC
    g = torch.Generator().manual_seed(110)
    b, s = 24, 4
    base = torch.rand(1, b, 121, 12, generator=g)
    src = (base * (0.6 + 0.8 * torch.rand(s, b, 1, 12, generator=g)) + 0.25 * torch.rand(s, b, 121, 12, generator=g))
    src[1, :6] = torch.rand(6, 121, 12, generator=g)            # uncorrelated patches: ncc near 1
    src[2, 3] = base[0, 3] * 2.0 + 0.1                           # perfectly correlated: cc = 1
    src[:, 5] = 0.37                                             # constant patches: zero variance -> the 1e-5 guard
    ref = base.clone().requires_grad_(True)
    src = src.clone().requires_grad_(True)
    ncc = compute_LNCC(ref, src)
    cot = torch.rand(ncc.shape, generator=g)
    g_ref, g_src = torch.autograd.grad((ncc * cot).sum(), [ref, src])
    npz("g11_lncc", ref=ref, src=src, ncc=ncc, cot=cot, g_ref=g_ref, g_src=g_src)


def g19_loss():
    """Loss.forward (models/losses/loss.py:24-93) on random predictions / targets: every returned term and the gradient of `loss` with
    respect to every differentiable prediction; two cases: the training configuration with all optional targets (confs/gens.conf:47-59), and
    the fine-tune configuration without pseudo points / depth targets (confs/gens_finetune.conf:32-41)."""
    from models.losses.loss import Loss
    out = {}
    for tag, conf, extras in (("a", dict(color_weight=1.0, sparse_weight=0.02, igr_weight=0.1, sparse_scale_factor=100, mfc_weight=1.0,
                                         smooth_weight=0.0001, tv_weight=0.0001, depth_weight=0.0, pseudo_sdf_weight=1.0, normal_weight=0.0,
                                         pseudo_depth_weight=0.05), True),
                              ("b", dict(color_weight=1.0, sparse_weight=0.0, igr_weight=0.1, sparse_scale_factor=100, mfc_weight=1.0,
                                         smooth_weight=0.0005, tv_weight=0.0001, pseudo_sdf_weight=1.0), False)):
        g = torch.Generator().manual_seed(190 + len(out))
        b, s = 40, 2 if tag == "b" else 4
        base = torch.rand(1, b, 121, 12, generator=g)
        preds = {
            "color_fine": torch.rand(b, 3, generator=g), "valid_mask": torch.rand(b, 1, generator=g) > 0.25,
            "gradient_error": torch.rand((), generator=g), "smooth_error": torch.rand((), generator=g), "tv_reg": torch.rand((), generator=g),
            "sparse_sdf": 0.02 * torch.randn(1024 + b * 128, 1, generator=g), "mid_inside_sphere": (torch.rand(b, 1, generator=g) > 0.3).float(),
            "ref_gray_val": base, "sampled_gray_val": base * (0.6 + 0.8 * torch.rand(s, b, 1, 12, generator=g)) + 0.25 * torch.rand(s, b, 121, 12, generator=g),
            "render_depth": 1.0 + torch.rand(b, generator=g),
        }
        targets = {"color": torch.rand(b, 3, generator=g)}
        if extras:
            preds["pseudo_sdf"] = 0.05 * torch.randn(300, 1, generator=g)
            pd = 1.0 + torch.rand(b, generator=g)
            pd[::5] = 0.0
            targets["pseudo_depth"] = pd
            dt = 1.0 + torch.rand(b, generator=g)
            dt[::3] = 0.0
            targets["depth"] = dt
        diff = [k for k in ("color_fine", "gradient_error", "smooth_error", "tv_reg", "sparse_sdf", "sampled_gray_val", "render_depth", "pseudo_sdf")
                if k in preds]
        for k in diff:
            preds[k] = preds[k].clone().requires_grad_(True)
        res = Loss(Conf(conf))(preds, targets)
        grads = torch.autograd.grad(res["loss"], [preds[k] for k in diff], allow_unused=True)
        for k, v in preds.items():
            out[f"{tag}.pred.{k}"] = v.detach()
        for k, v in targets.items():
            out[f"{tag}.target.{k}"] = v
        for k, v in res.items():
            out[f"{tag}.out.{k}"] = v.detach()
        for k, v in zip(diff, grads):
            out[f"{tag}.grad.{k}"] = torch.zeros_like(preds[k]) if v is None else v
        out[f"{tag}.conf"] = torch.tensor([conf.get(k, 0.0) for k in ("color_weight", "igr_weight", "sparse_weight", "mfc_weight", "smooth_weight",
                                                                       "tv_weight", "pseudo_sdf_weight", "pseudo_depth_weight", "sparse_scale_factor")])
    npz("g19_loss", **out)


def _install_cv2_stub():
    """cv2 is absent here: a stub provides the two calls the datasets make -- INTER_NEAREST resize (OpenCV's documented index
    rule) and decomposeProjectionMatrix (scipy.linalg.rq + the null vector of P by SVD; the product code uses a different
    route, numpy QR + a linear solve)."""
    import scipy.linalg
    cv2 = types.ModuleType("cv2")
    cv2.INTER_NEAREST = 0

    def resize(img, dsize, fx=None, fy=None, interpolation=0):
        w, h = dsize
        sh, sw = img.shape[:2]
        ys = np.minimum(np.floor(np.arange(h) * (float(sh) / h)).astype(int), sh - 1)
        xs = np.minimum(np.floor(np.arange(w) * (float(sw) / w)).astype(int), sw - 1)
        return img[ys[:, None], xs[None, :]]

    def decomposeProjectionMatrix(P):
        Pd = np.asarray(P, dtype=np.float64)
        K, R = scipy.linalg.rq(Pd[:, :3])
        D = np.diag(np.sign(np.diag(K)))
        K, R = K @ D, D @ R
        c = np.linalg.svd(Pd)[2][-1]
        return K.astype(P.dtype), R.astype(P.dtype), c.reshape(4, 1).astype(P.dtype), None, None, None, None

    cv2.resize, cv2.decomposeProjectionMatrix = resize, decomposeProjectionMatrix
    sys.modules["cv2"] = cv2


class _Conf(dict):
    def get_int(self, k, default=None): return int(self.get(k, default))
    def get_float(self, k, default=None): return float(self.get(k, default))
    def get_string(self, k, default=None): return self.get(k, default)
    def get_list(self, k, default=None): return self.get(k, default)


def _dump_item(out, prefix, item):
    for k, v in item.items():
        if isinstance(v, torch.Tensor):
            out[f"{prefix}.{k}"] = v.numpy()
        elif isinstance(v, (int, np.integer)):
            out[f"{prefix}.{k}"] = np.int64(v)
        elif isinstance(v, list):
            out[f"{prefix}.{k}"] = np.array(v, dtype=np.int64)
        elif isinstance(v, str):
            out[f"{prefix}.{k}"] = np.array(v)


def g12_dtu_dataset():
    """The reference's DTUDataset / DTUDatasetFinetune (datasets/dtu.py, dtu_finetune.py) on the synthetic tree of tests/dtu_fixture.py:
    one val and one train item, and the fine-tune dataset's three accessors (cv2 stubbed, see _install_cv2_stub)."""
    import random
    import tempfile
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import dtu_fixture
    _install_cv2_stub()
    from datasets.dtu import DTUDataset
    out = {}
    with tempfile.TemporaryDirectory() as root:
        dtu_fixture.make_dtu_tree(root)
        for mode, idx in (("val", 1), ("train", 0)):
            ds = DTUDataset(_Conf(dtu_fixture.conf_values(root, mode)), mode)
            random.seed(5)
            np.random.seed(6)
            torch.manual_seed(7)
            item = ds[idx]
            out[f"{mode}_len"] = len(ds)
            _dump_item(out, mode, item)
        from datasets.dtu_finetune import DTUDatasetFinetune
        torch.manual_seed(11)
        ft = DTUDatasetFinetune(_Conf(dtu_fixture.finetune_conf_values(root)), "finetune")
        items = {"all": ft.get_all_images(), "rand": ft.get_random_rays(torch.tensor(1)), "at": ft.get_rays_at(2)}
        out["ft.pseudo_ptses"] = ft.pseudo_ptses.numpy()
        out["ft.scale_mat"] = ft.scale_mat.numpy()
        for name, item in items.items():
            _dump_item(out, f"ft.{name}", item)
    npz("g12_dtu_dataset", **out)


def g13_bmvs_dataset():
    """The reference's BMVSDataset / BMVSDatasetFinetune (datasets/bmvs.py, bmvs_finetune.py) on the synthetic tree of
    tests/bmvs_fixture.py: one val and one train item, and the fine-tune dataset's three accessors."""
    import random
    import tempfile
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import bmvs_fixture
    _install_cv2_stub()
    from datasets.bmvs import BMVSDataset
    out = {}
    with tempfile.TemporaryDirectory() as root:
        bmvs_fixture.make_bmvs_tree(root)
        for mode, idx in (("val", 1), ("train", 0)):
            ds = BMVSDataset(_Conf(bmvs_fixture.conf_values(root, mode)), mode)
            random.seed(5)
            np.random.seed(6)
            torch.manual_seed(7)
            item = ds[idx]
            out[f"{mode}_len"] = len(ds)
            _dump_item(out, mode, item)
        from datasets.bmvs_finetune import BMVSDatasetFinetune
        torch.manual_seed(11)
        ft = BMVSDatasetFinetune(_Conf(bmvs_fixture.finetune_conf_values(root)), "finetune")
        items = {"all": ft.get_all_images(), "rand": ft.get_random_rays(torch.tensor(1)), "at": ft.get_rays_at(2)}
        out["ft.scale_mat"] = ft.scale_mat.numpy()
        out["ft.masks"] = ft.masks.numpy()
        for name, item in items.items():
            _dump_item(out, f"ft.{name}", item)
    npz("g13_bmvs_dataset", **out)


def g14_clean_mesh():
    """The reference's clean_mesh_by_mask (utils/clean_mesh.py:9-35) on a lattice mesh seen by the synthetic cameras.  Its module
    imports skimage / trimesh / open3d, none of which the function uses: empty stubs; the mesh is a stand-in with the three
    members the function touches (vertices, faces, update_faces)."""
    for name in ("skimage", "skimage.morphology", "trimesh", "open3d"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["skimage"].morphology = sys.modules["skimage.morphology"]
    from utils.clean_mesh import clean_mesh_by_mask
    sys.path.insert(0, REPO)
    from gens_amd import synthetic
    intrs, c2ws, _, _ = synthetic.make_cameras(5, 48, 64)
    g = torch.Generator().manual_seed(140)
    n = 14
    lin = torch.linspace(-1.2, 1.2, n)
    vx, vy = torch.meshgrid(lin, lin, indexing="ij")
    verts = torch.stack([vx, vy, 0.4 * torch.sin(3 * vx) * torch.cos(2 * vy)], -1).reshape(-1, 3) + 0.01 * torch.randn(n * n, 3, generator=g)
    idx = torch.arange(n * n).reshape(n, n)
    faces = torch.cat([torch.stack([idx[:-1, :-1], idx[1:, :-1], idx[:-1, 1:]], -1).reshape(-1, 3),
                       torch.stack([idx[1:, :-1], idx[1:, 1:], idx[:-1, 1:]], -1).reshape(-1, 3)]).numpy()
    yy, xx = torch.meshgrid(torch.arange(48.0), torch.arange(64.0), indexing="ij")
    masks = torch.stack([(((xx - 32 - 4 * v) / 22) ** 2 + ((yy - 24) / 17) ** 2 < 1).float() for v in range(5)])

    class Mesh:
        def __init__(self):
            self.vertices, self.faces = verts.double().numpy(), faces.copy()

        def update_faces(self, keep):
            self.faces = self.faces[keep]

    out = {"vertices": verts.double().numpy(), "faces": faces, "masks": masks, "intrs": intrs, "c2ws": c2ws}
    for nb in (1, 2):
        out[f"kept{nb}"] = clean_mesh_by_mask(Mesh(), masks, intrs, c2ws, min_nb_visible=nb).faces
    npz("g14_clean_mesh", **out)


def g16_backbones():
    """The reference's RegNetwork (models/modules/reg_network.py:105-169) and FeatureNetwork (feature_network_mnasnet.py:53-103) with
    seeded random weights.  RegNetwork imports nothing but torch; FeatureNetwork takes its trunk from torchvision (absent): the stub's
    `mnasnet1_0` hands it gens_amd's restatement of the MnasNet-1.0 `layers`, so this golden pins the reference's wiring around the
    trunk (stage cuts, decoder, heads, parameter names), not the trunk against torchvision."""
    import torch.nn as nn
    from models.modules.reg_network import RegNetwork
    out = {}
    torch.manual_seed(160)
    net = RegNetwork(_Conf(d_voluem=[8, 8, 8], d_out=[4, 4, 4], d_base=8)).eval()
    g = torch.Generator().manual_seed(161)
    vols = [torch.randn(1, 8, d, d, d, generator=g).requires_grad_(True) for d in (16, 8, 4)]
    cots = [torch.randn(1, 4, d, d, d, generator=g) for d in (16, 8, 4)]
    outs = net(vols)
    sum((o * c).sum() for o, c in zip(outs, cots)).backward()
    for i in range(3):
        out[f"reg.in{i}"], out[f"reg.cot{i}"], out[f"reg.out{i}"], out[f"reg.gin{i}"] = vols[i].detach(), cots[i], outs[i].detach(), vols[i].grad
    out["reg.keys"] = np.array(list(net.state_dict().keys()))
    for k, v in net.state_dict().items():
        out[f"reg.w.{k}"] = v
    out["reg.gw.conv0"] = net.conv0.conv.weight.grad

    from gens_amd.models.modules.feature_network import _mnasnet_trunk
    sys.modules["torchvision.models"].mnasnet1_0 = lambda pretrained=True: types.SimpleNamespace(
        layers=nn.Sequential(*_mnasnet_trunk(), nn.Identity(), nn.Identity(), nn.Identity()))
    from models.modules.feature_network_mnasnet import FeatureNetwork
    torch.manual_seed(162)
    fnet = FeatureNetwork(_Conf(d_out=[4, 4, 4, 4, 4]))
    imgs = torch.rand(2, 3, 64, 96, generator=g)
    out["feat.imgs"] = imgs
    with torch.no_grad():
        for i, o in enumerate(fnet.eval()(imgs)):
            out[f"feat.eval{i}"] = o
        for i, o in enumerate(fnet.train()(imgs)):            # BatchNorm on batch statistics; updates the running ones
            out[f"feat.train{i}"] = o
    sd = fnet.state_dict()
    out["feat.keys"] = np.array(list(sd.keys()))
    out["feat.shapes"] = np.array([str(tuple(v.shape)) for v in sd.values()])
    out["feat.sums"] = np.array([float(v.double().sum()) for v in sd.values()])
    out["feat.abs_sums"] = np.array([float(v.double().abs().sum()) for v in sd.values()])
    npz("g16_backbones", **out)


def g17_gens_forward(dims=(16, 8, 4), tag="g17_gens_forward", nv=3, seed=170):
    """The reference's WHOLE model, models/gens.py:12-157 `GenS.forward("train", ...)`: its FeatureNetwork (trunk: see g16), Volume,
    RegNetwork and ImplicitSurface on the CPU, one step with a scalar loss and its backward.  Backbone weights are the seeded
    initialisation (checksums stored); the implicit-surface weights are stored in full."""
    import torch.nn as nn
    from gens_amd.config import gens_model_conf
    from gens_amd.models.modules.feature_network import _mnasnet_trunk
    sys.modules["torchvision.models"].mnasnet1_0 = lambda pretrained=True: types.SimpleNamespace(
        layers=nn.Sequential(*_mnasnet_trunk(), nn.Identity(), nn.Identity(), nn.Identity()))
    from models.gens import GenS
    torch.manual_seed(seed)
    model = GenS(Conf(dict(gens_model_conf(volume_dims=dims)))).train()
    h, w, n_rays = 64, 96, 16
    sc = synthetic.make_scene(nv=nv, h=h, w=w, n_levels=1, seed=seed + 1)
    g = torch.Generator().manual_seed(seed + 2)
    pix = torch.stack([torch.randint(8, w - 8, (n_rays,), generator=g), torch.randint(8, h - 8, (n_rays,), generator=g)], -1)
    rays_o, rays_d = synthetic.make_rays(sc["intrs"], sc["c2ws"], h, w, pixels=pix)
    ipts = {"imgs": sc["imgs"], "intrs": sc["intrs"], "c2ws": sc["c2ws"], "rays_o": rays_o, "rays_d": rays_d, "near": sc["near"], "far": sc["far"],
            "pseudo_pts": torch.rand(64, 3, generator=g) - 0.5}
    d = {"in." + k: v for k, v in ipts.items()}
    d["pix"] = pix
    for k, v in model.implicit_surface.state_dict().items():
        d["sd." + k] = v
    sd = model.state_dict()
    names = [k for k in sd if not k.startswith("implicit_surface.")]
    d["backbone.keys"] = np.array(names)
    d["backbone.sums"] = np.array([float(sd[k].double().sum()) for k in names])
    d["backbone.abs_sums"] = np.array([float(sd[k].double().abs().sum()) for k in names])
    torch.manual_seed(seed + 3)
    out = model("train", ipts, cos_anneal_ratio=0.7, step=3)
    hit = out["mid_inside_sphere"].reshape(1, -1, 1, 1)
    loss = (out["color_fine"].abs().sum() + 0.1 * out["gradient_error"] + 0.01 * out["smooth_error"] + 0.01 * out["tv_reg"]
            + torch.exp(-out["sparse_sdf"].abs() * 100).mean() + (((out["sampled_gray_val"] - out["ref_gray_val"]) ** 2) * hit).mean()
            + 0.1 * out["render_depth"].sum() + out["pseudo_sdf"].abs().mean())
    loss.backward()
    for k, v in out.items():
        if isinstance(v, torch.Tensor):
            d["out." + k] = v
    d["loss"] = loss
    params = dict(model.named_parameters())
    for k in ("feature_network.layer1.0.weight", "feature_network.out_layer1.weight", "feature_network.out_layer5.weight",
              "reg_network.conv0.conv.weight", "reg_network.out_layers.0.bias", "reg_network.decoder_layers.2.conv.weight",
              "implicit_surface.sdf_network.lin0.weight_v", "implicit_surface.sdf_network.lin6.bias", "implicit_surface.deviation_network.variance"):
        d["grad." + k] = params[k].grad
    for k, p in params.items():             # every implicit-surface gradient (the fused training kernels produce all of them)
        if k.startswith("implicit_surface.") and p.grad is not None:
            d["grad." + k] = p.grad
    d["dims"] = np.array(dims)
    npz(tag, **d)


def g18_gens_finetune(dims=(16, 8, 4), tag="g18_gens_finetune", seed=180):
    """The reference's per-scene fine-tune path, models/gens.py:63-85,141-155: `init_volumes` (CNN outputs frozen into parameters) on four
    views, then `forward("finetune", ...)` on a re-ordered subset of them (view_ids), loss and backward into the volume parameters."""
    import torch.nn as nn
    from gens_amd.config import gens_model_conf
    from gens_amd.models.modules.feature_network import _mnasnet_trunk
    sys.modules["torchvision.models"].mnasnet1_0 = lambda pretrained=True: types.SimpleNamespace(
        layers=nn.Sequential(*_mnasnet_trunk(), nn.Identity(), nn.Identity(), nn.Identity()))
    from models.gens import GenS
    torch.manual_seed(seed)
    model = GenS(Conf(dict(gens_model_conf(volume_dims=dims)))).train()
    h, w, nv, n_rays = 64, 96, 4, 16
    nl = len(dims)
    sc = synthetic.make_scene(nv=nv, h=h, w=w, n_levels=1, seed=seed + 1)
    d = {"all.imgs": sc["imgs"], "all.intrs": sc["intrs"], "all.c2ws": sc["c2ws"]}
    for k, v in model.implicit_surface.state_dict().items():
        d["sd." + k] = v
    model.init_volumes({"imgs": sc["imgs"], "intrs": sc["intrs"], "c2ws": sc["c2ws"]})
    for i in range(nl):
        d[f"init.volume{i}"], d[f"init.mask{i}"] = model.volumes[i].detach().clone(), model.mask_volmes[i].detach().clone()
        if dims[i] > 32:      # keep the fixture small: every second voxel per axis of the values, the mask as bits
            d[f"init.volume{i}"] = d[f"init.volume{i}"][..., ::2, ::2, ::2].contiguous()
            d[f"init.mask{i}"] = np.packbits(d[f"init.mask{i}"].numpy().astype(np.uint8).reshape(-1))
    for i in range(5):
        d[f"init.feature{i}"] = model.features[i].detach().clone()
    view_ids = [2, 0, 3]
    g = torch.Generator().manual_seed(seed + 2)
    pix = torch.stack([torch.randint(8, w - 8, (n_rays,), generator=g), torch.randint(8, h - 8, (n_rays,), generator=g)], -1)
    intrs, c2ws = sc["intrs"][view_ids], sc["c2ws"][view_ids]
    rays_o, rays_d = synthetic.make_rays(intrs, c2ws, h, w, pixels=pix)
    ipts = {"imgs": sc["imgs"][view_ids], "intrs": intrs, "c2ws": c2ws, "rays_o": rays_o, "rays_d": rays_d, "near": sc["near"], "far": sc["far"],
            "pseudo_pts": torch.rand(64, 3, generator=g) - 0.5, "view_ids": view_ids}
    for k, v in ipts.items():
        d["in." + k] = np.array(v) if k == "view_ids" else v
    torch.manual_seed(seed + 3)
    out = model("finetune", ipts, cos_anneal_ratio=1.0, step=11)
    hit = out["mid_inside_sphere"].reshape(1, -1, 1, 1)
    loss = (out["color_fine"].abs().sum() + 0.1 * out["gradient_error"] + 0.01 * out["smooth_error"] + 0.01 * out["tv_reg"]
            + torch.exp(-out["sparse_sdf"].abs() * 100).mean() + (((out["sampled_gray_val"] - out["ref_gray_val"]) ** 2) * hit).mean()
            + 0.1 * out["render_depth"].sum() + out["pseudo_sdf"].abs().mean())
    loss.backward()
    for k, v in out.items():
        if isinstance(v, torch.Tensor):
            d["out." + k] = v
    d["loss"] = loss
    for i in range(nl):
        d[f"grad.volume{i}"] = model.volumes[i].grad
        if dims[i] > 32:      # every second voxel per axis, like the values
            d[f"grad.volume{i}"] = model.volumes[i].grad[..., ::2, ::2, ::2].contiguous()
    d["grad.lin0"] = model.implicit_surface.sdf_network.lin0.weight_v.grad
    for k, p in model.implicit_surface.named_parameters():
        if p.grad is not None:
            d["grad.implicit_surface." + k] = p.grad
    d["dims"] = np.array(dims)
    npz(tag, **d)


def main():
    _install_shims()
    if len(sys.argv) > 1 and sys.argv[1] == "l5":            # the shipped level count (confs/gens.conf:63-67,86): round-2 goldens
        from models.modules.volume import Volume
        from models.modules import implicit_surface as isurf_mod
        surf, sc, vols, masks = g9_render(isurf_mod, Volume, "g9c_render_l5", seed=120, cos_anneal=1.0, step=7, n_rays=16, variance=0.55,
                                          nv=5, dims=(24, 16, 8, 6, 4))
        g15_validate(surf, sc, vols, masks, tag="g15b_validate_l5")
        g9d_config0(isurf_mod, Volume)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "g1c":
        from models.modules.volume import Volume
        g1c_volume_tiles(Volume)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "g17b":
        g17_gens_forward(dims=(64, 32, 16, 8, 4), tag="g17b_gens_forward_l5", nv=4, seed=270)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "g18b":
        g18_gens_finetune(dims=(64, 32, 16, 8, 4), tag="g18b_gens_finetune_l5", seed=280)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "g18":
        g18_gens_finetune()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "g17":
        g17_gens_forward()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "g16":
        g16_backbones()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "g12":
        g12_dtu_dataset()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "g13":
        g13_bmvs_dataset()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "g14":
        g14_clean_mesh()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "g15":           # the validate golden (needs the model of g9b)
        from models.modules.volume import Volume
        from models.modules import implicit_surface as isurf_mod
        surf, sc, vols, masks = g9_render(isurf_mod, Volume, "g9b_render", seed=95, cos_anneal=1.0, step=7, n_rays=16, variance=0.55)
        g15_validate(surf, sc, vols, masks)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "g11":           # regenerate only the loss golden
        g11_lncc()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "g19":           # the Loss module
        g19_loss()
        return
    from models.modules.volume import Volume
    from models.modules import projector
    from models.modules import implicit_surface as isurf_mod

    g1_volume(Volume)
    g1c_volume_tiles(Volume)
    g2_lookup(projector)
    g3_nearest(projector)
    g4_feature(projector)
    g5_upsample(isurf_mod, projector)
    g7_patchwarp(projector)
    g8_tv(isurf_mod)
    g9_render(isurf_mod, Volume, "g9a_render", seed=90, cos_anneal=0.5, step=None, n_rays=24)
    surf, sc, vols, masks = g9_render(isurf_mod, Volume, "g9b_render", seed=95, cos_anneal=1.0, step=7, n_rays=16, variance=0.55)
    g10_geometry(surf, vols)
    g15_validate(surf, sc, vols, masks)
    g11_lncc()
    g19_loss()
    g12_dtu_dataset()
    g13_bmvs_dataset()
    g14_clean_mesh()
    leaked = [p for p, _, fs in os.walk(REF) for f in fs if f.endswith(".pyc")]
    assert not leaked, f"bytecode leaked into the reference tree: {leaked}"


if __name__ == "__main__":
    main()
