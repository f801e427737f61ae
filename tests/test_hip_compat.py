"""GPU tests of the reference-convention boundary (gens_amd/compat/cuda_gridsample.py) and of view counts above eight."""
import pytest
import torch

from oracle import gens_oracle as K

pytestmark = pytest.mark.gpu


def close(a, b, atol, what):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert (a - b).abs().max() <= atol, f"{what}: max err {(a - b).abs().max():.3e}"


def test_grid_sample_3d_in_the_references_calling_convention(golden):
    """`cug.grid_sample_3d(input (1,C,D,H,W), grid (1,1,1,N,3), 'zeros', True)` exactly as projector.py:223-229 calls it, value + first +
    second derivatives against golden g2 (the reference's own Function pair), one level per call like the reference's loop."""
    from gens_amd.compat import cuda_gridsample as cug
    g = golden("g2_lookup")
    pts = g["pts"].cuda().requires_grad_(True)
    vols = [g[f"vol{i}"].cuda().requires_grad_(True) for i in range(3)]
    x = pts.unsqueeze(0).unsqueeze(0).unsqueeze(0).flip(dims=[-1])                     # projector.py:223
    feats = torch.cat([cug.grid_sample_3d(v, x, padding_mode="zeros", align_corners=True).reshape(-1, pts.shape[0]).permute(1, 0)
                       for v in vols], -1)                                               # projector.py:229,232-233
    close(feats, g["feats"], 1e-5, "forward")
    go = g["gO"].cuda().requires_grad_(True)
    grads = torch.autograd.grad(feats, [pts] + vols, go, create_graph=True)
    close(grads[0], g["gP"], 2e-5, "gP")
    for i in range(3):
        close(grads[1 + i], g[f"gV{i}"], 2e-5, f"gV{i}")
    outs = torch.autograd.grad(grads[0], [go, pts] + vols, g["ggG"].cuda())
    close(outs[0], g["ggO"], 5e-5, "ggO")
    close(outs[1], g["gP2"], 3e-4, "gP2")
    for i in range(3):
        close(outs[2 + i], g[f"gV2_{i}"], 5e-5, f"gV2_{i}")


def test_grad2_3d_entry_point_matches_the_oracle():
    """`gridsample_grad2.grad2_3d(ggI, ggG, gO, input, grid, padding_mode, align_corners) -> [ggO, gI, gG]` (gridsample_cuda.cpp:42-56)
    in the reference's tensor layouts, including the grad2_grad_input branch, against the CPU oracle."""
    from gens_amd.compat import cuda_gridsample as cug
    gen = torch.Generator().manual_seed(3)
    d, n = 7, 200
    vol = torch.randn(1, 4, d, d, d, generator=gen)
    pts = torch.rand(n, 3, generator=gen) * 2.4 - 1.2
    go = torch.randn(n, 4, generator=gen)
    ggp = torch.randn(n, 3, generator=gen)
    ggv = torch.randn(1, 4, d, d, d, generator=gen)
    ref_ggo, ref_gv, ref_gp = K.lookup_volume_bwd2([ggv], ggp, go, [vol], pts)
    grid = pts.flip(-1).reshape(1, 1, 1, n, 3).cuda()
    out = cug.grad2_3d(ggv.cuda(), ggp.flip(-1).reshape(1, 1, 1, n, 3).cuda(), go.t().reshape(1, 4, 1, 1, n).cuda(), vol.cuda(), grid, False, True)
    close(out[0].reshape(4, n).t(), ref_ggo, 5e-5, "ggO")
    close(out[1], ref_gv[0], 5e-5, "gI")
    close(out[2].reshape(n, 3).flip(-1), ref_gp, 3e-4, "gG")
    # 'border' (never passed by the reference's own callers, projector.py:229,238) runs on the general kernels: tests/test_grid_sample_general.py
    assert cug.grid_sample_3d(vol.cuda(), grid, padding_mode="border").shape == (1, 4, 1, 1, n)


def test_more_than_eight_views():
    """The reference's fine-tune sets hold up to 11 views (dtu_finetune.py: ref + 10 sources): K1 (generic kernel above 8 views), K4 and
    the fused blend kernel with ten views against the oracle."""
    from gens_amd import ops, synthetic
    from gens_amd.models.modules.blending_network import BlendingNetwork
    from oracle import render_oracle as R
    nv = 10
    sc = synthetic.make_scene(nv=7, h=48, w=64, n_levels=5, seed=5)
    g = torch.Generator().manual_seed(6)
    # ten views: the seven synthetic cameras plus three of them nudged
    c2ws = torch.cat([sc["c2ws"], sc["c2ws"][1:4].clone()], 0)
    c2ws[7:, :3, 3] += 0.05
    intrs = torch.cat([sc["intrs"], sc["intrs"][1:4]], 0)
    imgs = torch.rand(nv, 3, 48, 64, generator=g)
    feats = [torch.randn(nv, 4, 48 >> i, 64 >> i, generator=g) for i in range(5)]
    dims = [16, 8]
    ref_v, ref_m = K.volume_build(feats[:2], intrs, c2ws, dims)
    v, m = ops.volume_build([f.cuda() for f in feats[:2]], intrs.cuda(), c2ws.cuda(), dims)
    for i in range(2):
        assert (m[i].cpu() != ref_m[i]).float().mean() <= 1e-3
        close(v[i] * (m[i] == ref_m[i].cuda()), ref_v[i] * (m[i].cpu() == ref_m[i]), 1e-4, f"volume{i}")
    pts = (torch.rand(200, 3, generator=g) * 1.6 - 0.8)
    views = ops.SceneViews(imgs.cuda(), intrs.cuda(), c2ws.cuda(), [f.cuda() for f in feats])
    fv, rd, mk = ops.lookup_feature(pts.cuda(), views)
    rfv, rrd, rmk = K.lookup_feature(pts, imgs, intrs, c2ws, feats)
    assert torch.equal(mk.cpu(), rmk)
    close(torch.nan_to_num(fv) * mk[..., None], torch.nan_to_num(rfv) * rmk[..., None], 1e-4, "feat_views")      # white-noise maps: O(1) / px gradients x 1e-5 px of float32 projection round-off
    torch.manual_seed(1)
    net = BlendingNetwork(d_feature=20).cuda()
    sd = {"color_network." + k: t.detach().cpu() for k, t in net.state_dict().items()}
    ref = R.blend_mlp(sd, torch.nan_to_num(rfv), torch.nan_to_num(rrd), rmk)
    rgb, vis = ops.blend_views(ops.BlendPlan(net), views, pts.cuda())
    live = rmk.any(1)
    assert torch.equal(vis.bool().cpu(), rmk)
    close(rgb.cpu()[live], ref[live], 1e-4, "blend, S = 9")
