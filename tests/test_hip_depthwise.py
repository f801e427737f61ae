"""K21 (gens_depthwise_conv2d_*): the depth-wise convolutions of the MnasNet trunk (feature_network_mnasnet.py:53-103 -> torchvision's
MNASNet: nn.Conv2d(c, c, k, padding=k//2, stride=s, groups=c, bias=False)) against ATen's convolution in float64 on the CPU -- value, data
gradient and weight gradient for every (k, stride) the trunk uses, at wide, narrow, odd and one-pixel planes -- and the whole FeatureNetwork with
the kernels on and off."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

WORST, REL_L2 = 2e-3, 1e-3
SHAPES = [(2, 5, 37, 70), (1, 3, 15, 20), (3, 8, 64, 129), (2, 4, 1, 1), (1, 2, 2, 3), (1, 6, 30, 40),
          (5, 1152, 5, 6), (5, 576, 10, 12), (2, 1200, 15, 20)]          # the deepest stage's planes: many channels, a few dozen pixels


@pytest.mark.parametrize("k", [3, 5])
@pytest.mark.parametrize("stride", [1, 2])
def test_depthwise_matches_aten_float64(k, stride):
    from gens_amd import ops
    for i, (n, c, h, w) in enumerate(SHAPES):
        g = torch.Generator().manual_seed(100 * k + 10 * stride + i)
        x = torch.randn(n, c, h, w, generator=g)
        wt = torch.randn(c, 1, k, k, generator=g)
        xr, wr = x.double().requires_grad_(True), wt.double().requires_grad_(True)
        ref = F.conv2d(xr, wr, None, stride, k // 2, 1, c)
        go = torch.randn(ref.shape, generator=g)
        gx_ref, gw_ref = torch.autograd.grad(ref, [xr, wr], go.double())
        xd, wd = x.cuda().requires_grad_(True), wt.cuda().requires_grad_(True)
        assert ops.depthwise_supported(xd, wd, None, (stride, stride), (k // 2, k // 2), (1, 1), c)
        out = ops.depthwise_conv2d(xd, wd, stride)
        assert out.shape == ref.shape
        gx, gw = torch.autograd.grad(out, [xd, wd], go.cuda())
        for name, a, b in (("value", out, ref), ("data gradient", gx, gx_ref), ("weight gradient", gw, gw_ref)):
            err = float((a.detach().cpu().double() - b.detach()).abs().max())
            scale = max(1.0, float(b.abs().max()))
            assert err <= 2e-5 * scale, (name, (n, c, h, w), err, scale)


def test_depthwise_declines_what_it_does_not_cover():
    from gens_amd import ops
    x, w3 = torch.randn(1, 4, 8, 8).cuda(), torch.randn(4, 1, 3, 3).cuda()
    assert not ops.depthwise_supported(x, w3, None, (1, 1), (0, 0), (1, 1), 4)                  # padding != k // 2
    assert not ops.depthwise_supported(x, w3, None, (1, 1), (1, 1), (2, 2), 4)                  # dilation
    assert not ops.depthwise_supported(x, w3, torch.zeros(4).cuda(), (1, 1), (1, 1), (1, 1), 4)  # bias
    assert not ops.depthwise_supported(x, torch.randn(4, 2, 3, 3).cuda(), None, (1, 1), (1, 1), (1, 1), 2)   # grouped, not depth-wise
    assert not ops.depthwise_supported(x.cpu(), w3.cpu(), None, (1, 1), (1, 1), (1, 1), 4)      # CPU tensors stay on PyTorch's convolution
    assert not ops.depthwise_supported(x, torch.randn(4, 1, 7, 7).cuda(), None, (1, 1), (3, 3), (1, 1), 4)


def test_feature_network_with_and_without_the_depthwise_kernels():
    """The whole 2-D CNN of GenS (MnasNet trunk + decoder), forward and backward, with DepthwiseConv2d on K21 and on MIOpen, against the same
    network in float64 on the CPU.  BatchNorm runs on its running statistics here (eval mode): with BATCH statistics over the few dozen
    elements of the 1/32 level the float32 network is chaotic in its deepest stage -- MIOpen's own gradients are 12 - 16 % off the float64
    ones there -- and says nothing about a convolution kernel."""
    import copy
    from gens_amd.config import gens_model_conf
    from gens_amd.models.modules import feature_network as FN
    torch.manual_seed(0)
    net = FN.FeatureNetwork(gens_model_conf()["feature_network"]).cuda()
    n_dw = sum(isinstance(m, FN.DepthwiseConv2d) for m in net.modules())
    assert n_dw == 17                                       # the stem's + one per inverted residual (3 + 3 + 3 + 2 + 4 + 1)
    imgs = torch.rand(5, 3, 160, 192).cuda()
    net.train()
    with torch.no_grad():                                   # running statistics from a few batches, then frozen
        for _ in range(3):
            net(torch.rand(5, 3, 160, 192).cuda())
    net.eval()
    g = torch.Generator().manual_seed(1)
    cots, res = None, {}
    for on in (True, False):
        FN.DepthwiseConv2d.use_k21 = on
        try:
            for p in net.parameters():
                p.grad = None
            feats = net(imgs)
            if cots is None:
                cots = [torch.randn(f.shape, generator=g).cuda() for f in feats]
            sum((f * c).sum() for f, c in zip(feats, cots)).backward()
            res[on] = ([f.detach().clone() for f in feats], {k: p.grad.detach().clone() for k, p in net.named_parameters()})
        finally:
            FN.DepthwiseConv2d.use_k21 = True
    ref_net = copy.deepcopy(net).double().cpu()
    ref_feats = ref_net(imgs.double().cpu())
    sum((f * c.double().cpu()).sum() for f, c in zip(ref_feats, cots)).backward()
    ref = {k: p.grad for k, p in ref_net.named_parameters()}
    for on in (True, False):
        for a, b in zip(res[on][0], ref_feats):
            assert float((a.double().cpu() - b.detach()).abs().max()) <= 1e-4 * max(1.0, float(b.abs().max())), on
    top = max(float(v.abs().max()) for v in ref.values())
    err = {on: {k: float((res[on][1][k].double().cpu() - b).abs().max()) / max(float(b.abs().max()), 1e-3 * top) for k, b in ref.items()} for on in (True, False)}
    worst = sorted(err[True], key=lambda k: -err[True][k])[:6]
    print("gradient error against float64 (max |diff| / max |ref|), K21 | MIOpen:", {k: "%.1e | %.1e" % (err[True][k], err[False][k]) for k in worst})
    bad = {k: (err[True][k], err[False][k]) for k in ref if err[True][k] > 1e-3 and err[True][k] > 3.0 * err[False][k]}
    assert not bad, bad
