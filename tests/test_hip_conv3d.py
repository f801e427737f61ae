"""K15 (gens_amd/csrc/k15_conv3d.hip): the 3 x 3 x 3 convolutions of the cost-volume U-Net against torch's own float32 convolution
on the CPU (what the reference's nn.Conv3d / nn.ConvTranspose3d compute, reg_network.py:15,38) -- value, data gradient, weight
gradient, bias gradient.  Floating point with a different order of summation: tolerance 2e-5 of the tensor's largest magnitude."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _close(name, got, want, tol=2e-5):
    got, want = got.detach().cpu().double(), want.detach().double()
    assert got.shape == want.shape, (name, got.shape, want.shape)
    err = (got - want).abs().max().item() / max(want.abs().max().item(), 1e-12)
    assert err < tol, f"{name}: error {err:.2e} of the largest magnitude"


@pytest.mark.parametrize("cin,cout,dims,stride,bias", [
    (8, 8, (16, 16, 16), 1, False), (8, 4, (12, 8, 20), 1, True), (8, 8, (16, 16, 16), 2, False), (24, 32, (8, 8, 8), 2, False),
    (3, 5, (6, 10, 12), 1, True), (16, 16, (4, 4, 4), 1, False), (7, 13, (4, 6, 2), 2, True), (32, 32, (8, 4, 68), 1, False),
    (1, 1, (2, 2, 2), 1, False), (9, 3, (2, 2, 2), 2, False),
    (8, 8, (4, 6, 64), 1, True), (12, 5, (3, 3, 128), 1, False), (4, 4, (2, 5, 192), 1, False),       # z % 64 == 0: the LDS neighbour exchange
    (8, 8, (4, 4, 128), 2, True), (5, 12, (2, 6, 256), 2, False),                                       # coarse z % 64 == 0 at stride 2
    (8, 8, (9, 6, 64), 1, True), (6, 7, (5, 9, 128), 1, False), (3, 8, (17, 4, 64), 1, True),              # five to eight channels out, at most eight in: the matrix-core gather (forward and data gradient)
    (8, 3, (3, 11, 64), 1, True), (16, 4, (2, 17, 128), 1, False),                                      # four output channels or fewer: the weight gradient's two-rows-per-wave form (the U-Net's heads)
    (12, 24, (6, 10, 64), 2, False), (40, 20, (2, 14, 64), 2, True),                                    # coarse z = 32: one segment of the stride-2 matrix-core weight gradient, two blocks of P channels
    (72, 128, (2, 2, 2), 2, False), (128, 128, (2, 2, 2), 1, False), (64, 64, (4, 4, 4), 1, True), (40, 64, (4, 4, 4), 2, False)])   # the deep levels of the 5-stage U-Net
def test_conv3d_matches_torch(cin, cout, dims, stride, bias):
    from gens_amd import ops
    g = torch.Generator().manual_seed(cin * 100 + cout)
    x = torch.randn(1, cin, *dims, generator=g).requires_grad_(True)
    w = (torch.randn(cout, cin, 3, 3, 3, generator=g) / (27 * cin) ** 0.5).requires_grad_(True)
    b = torch.randn(cout, generator=g).requires_grad_(True) if bias else None
    y = F.conv3d(x, w, b, stride=stride, padding=1)
    cot = torch.randn(y.shape, generator=g)
    grads = torch.autograd.grad(y, [x, w] + ([b] if bias else []), cot)
    xd, wd = x.detach().cuda().requires_grad_(True), w.detach().cuda().requires_grad_(True)
    bd = b.detach().cuda().requires_grad_(True) if bias else None
    yd = ops.conv3d(xd, wd, bd, stride)
    _close("value", yd, y)
    gd = torch.autograd.grad(yd, [xd, wd] + ([bd] if bias else []), cot.cuda())
    for name, a, r in zip(("dgrad", "wgrad", "bgrad"), gd, grads):
        _close(name, a, r)


@pytest.mark.parametrize("cin,cout,dims", [(8, 8, (8, 8, 8)), (32, 16, (4, 4, 4)), (16, 8, (6, 10, 34)), (5, 3, (3, 2, 7)), (1, 9, (1, 1, 1)),
                                           (128, 64, (2, 2, 2)), (64, 32, (4, 4, 4)), (8, 8, (2, 3, 64)), (6, 10, (3, 2, 128)), (24, 40, (3, 5, 32))])
def test_conv_transpose3d_matches_torch(cin, cout, dims):
    from gens_amd import ops
    g = torch.Generator().manual_seed(cin * 100 + cout + 7)
    x = torch.randn(1, cin, *dims, generator=g).requires_grad_(True)
    w = (torch.randn(cin, cout, 3, 3, 3, generator=g) / (27 * cin) ** 0.5).requires_grad_(True)
    y = F.conv_transpose3d(x, w, stride=2, padding=1, output_padding=1)
    assert y.shape[2:] == tuple(2 * d for d in dims)
    cot = torch.randn(y.shape, generator=g)
    grads = torch.autograd.grad(y, [x, w], cot)
    xd, wd = x.detach().cuda().requires_grad_(True), w.detach().cuda().requires_grad_(True)
    yd = ops.conv_transpose3d(xd, wd)
    _close("value", yd, y)
    for name, a, r in zip(("dgrad", "wgrad"), torch.autograd.grad(yd, [xd, wd], cot.cuda()), grads):
        _close(name, a, r)


def test_conv3d_linearity_and_adjointness_at_full_size():
    """256^3, 8 -> 8 channels (the U-Net's first layer): <conv(x), y> = <x, conv^T(y)> ties the forward kernel to the data-gradient
    kernel, and <conv_w(x), y> = <w, wgrad(x, y)> ties it to the weight-gradient kernel; both hold to float32 summation error."""
    from gens_amd import ops
    g = torch.Generator(device="cuda").manual_seed(3)
    for stride in (1, 2):
        x = torch.randn(1, 8, 256, 256, 256, device="cuda", generator=g).requires_grad_(True)
        w = (torch.randn(8, 8, 3, 3, 3, device="cuda", generator=g) / 216 ** 0.5).requires_grad_(True)
        y = ops.conv3d(x, w, None, stride)
        cot = torch.randn(y.shape, device="cuda", generator=g)
        gx, gw = torch.autograd.grad(y, [x, w], cot)
        lhs = (y.double() * cot.double()).sum().item()
        assert abs(lhs - (x.double() * gx.double()).sum().item()) < 1e-6 * abs(lhs) + 1e-3 * y.numel() ** 0.5 * 1e-3
        assert abs(lhs - (w.double() * gw.double()).sum().item()) < 1e-5 * abs(lhs) + 1.0
        z = ops.conv3d(2.0 * x.detach(), w.detach(), None, stride)
        assert torch.equal(z, 2.0 * y.detach())                                            # scaling by 2 is exact in float32


def _unet_run(net, vols, cots):
    outs = net(vols)
    sum((o * c).sum() for o, c in zip(outs, cots)).backward()
    return [o.detach() for o in outs], [v.grad for v in vols], [p.grad for p in net.parameters()]


def _err(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-12)


@pytest.mark.parametrize("dims", [(32, 16, 8), (128, 64, 32, 16, 8)])       # three stages; the five of confs/gens.conf (up to 72 -> 128 channels)
def test_reg_network_on_the_device_matches_the_cpu_module(dims):
    """The U-Net on K15 / K16 against the same module run by torch on the CPU.  Its deep levels normalise over a handful of voxels
    (instance-norm over 4^3 at the bottom of five stages), which amplifies float32 round-off in the gradients: the yardstick is the
    module in float64, and the device result may be off by no more than a few times what torch's own float32 CPU run is off by."""
    import copy
    from gens_amd.config import Conf
    from gens_amd.models.modules.reg_network import RegNetwork
    torch.manual_seed(11)
    n = len(dims)
    net = RegNetwork(Conf({"d_voluem": [8] * n, "d_out": [4] * n, "d_base": 8}))
    g = torch.Generator().manual_seed(12)
    vols = [torch.randn(1, 8, d, d, d, generator=g) for d in dims]
    cots = [torch.randn(1, 4, d, d, d, generator=g) for d in dims]
    ref = _unet_run(copy.deepcopy(net).double(), [v.double().requires_grad_(True) for v in vols], [c.double() for c in cots])
    cpu = _unet_run(copy.deepcopy(net), [v.clone().requires_grad_(True) for v in vols], cots)
    dev = _unet_run(copy.deepcopy(net).cuda(), [v.cuda().requires_grad_(True) for v in vols], [c.cuda() for c in cots])
    names = [f"out{i}" for i in range(n)], [f"gin{i}" for i in range(n)], [k for k, _ in net.named_parameters()]
    for group, r_group, c_group, d_group, floor in zip(names, ref, cpu, dev, (1e-5, 2e-5, 1e-4)):
        for name, r, c, d in zip(group, r_group, c_group, d_group):
            e_cpu, e_dev = _err(c, r), _err(d, r)
            assert e_dev <= max(4.0 * e_cpu, floor), f"{name}: device {e_dev:.2e}, torch float32 on the CPU {e_cpu:.2e} (both against float64)"


def test_instnorm_relu_with_the_fused_skip_addition():
    from gens_amd import ops
    g = torch.Generator().manual_seed(21)
    x = torch.randn(1, 8, 8, 8, 64, generator=g).requires_grad_(True)
    skip = torch.randn(1, 8, 8, 8, 64, generator=g).requires_grad_(True)
    cot = torch.randn(1, 8, 8, 8, 64, generator=g)
    y = torch.relu(F.instance_norm(x.double(), eps=1e-5)) + skip.double()
    gx, gs = torch.autograd.grad(y, [x, skip], cot.double())
    xd, sd = x.detach().cuda().requires_grad_(True), skip.detach().cuda().requires_grad_(True)
    yd = ops.instnorm_relu(xd, 1e-5, sd)
    _close("value", yd, y, 1e-5)
    gxd, gsd = torch.autograd.grad(yd, [xd, sd], cot.cuda())
    assert torch.equal(gsd.cpu(), cot)                                               # the skip branch passes the gradient through
    keep = F.instance_norm(x.detach().double(), eps=1e-5).abs() > 1e-5
    assert ((gxd.cpu().double() - gx) * keep).abs().max().item() / gx.abs().max().item() < 2e-5


@pytest.mark.parametrize("shape", [(5, 8, 48, 64), (3, 96, 3, 5), (2, 16, 15, 20), (1, 24, 7, 9)])
def test_instnorm_relu_on_a_batch_of_planes_is_instance_norm_2d(shape):
    """nn.InstanceNorm2d (default form) + ReLU + the decoder's skip addition of the feature pyramid (feature_network_mnasnet.py:29-50, :97-100)
    on K16 as n * c planes, value and both gradients against float64 torch; sizes not a multiple of four take the scalar kernels."""
    from gens_amd import ops
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(*shape, generator=g).requires_grad_(True)
    skip = torch.randn(*shape, generator=g).requires_grad_(True)
    cot = torch.randn(*shape, generator=g)
    for with_skip in (True, False):
        y = torch.relu(F.instance_norm(x.double(), eps=1e-5)) + (skip.double() if with_skip else 0.0)
        gx = torch.autograd.grad(y, x, cot.double())[0]
        xd, sd = x.detach().cuda().requires_grad_(True), skip.detach().cuda().requires_grad_(True)
        yd = ops.instnorm_relu(xd, 1e-5, sd if with_skip else None)
        assert yd.shape == x.shape
        _close("value", yd, y, 1e-5)
        if with_skip:
            gxd, gsd = torch.autograd.grad(yd, [xd, sd], cot.cuda())
            assert torch.equal(gsd.cpu(), cot)
        else:
            gxd = torch.autograd.grad(yd, xd, cot.cuda())[0]
        keep = F.instance_norm(x.detach().double(), eps=1e-5).abs() > 1e-5
        assert ((gxd.cpu().double() - gx) * keep).abs().max().item() / gx.abs().max().item() < 5e-5


def test_feature_decoder_blocks_on_k16_match_the_torch_route(monkeypatch):
    """_Deconv2d (ConvTranspose2d -> InstanceNorm2d -> ReLU, + the encoder's map) with the norm on K16 against the same module on aten's
    instance-norm route (GENS_NO_K16_2D), forward and all gradients."""
    from gens_amd.models.modules import feature_network as fn
    torch.manual_seed(3)
    blk = fn._Deconv2d(24, 16).cuda()
    x = torch.randn(5, 24, 30, 40, device="cuda", requires_grad=True)
    skip = torch.randn(5, 16, 60, 80, device="cuda", requires_grad=True)
    cot = torch.randn(5, 16, 60, 80, device="cuda")
    res = {}
    for on in (True, False):
        monkeypatch.setattr(fn._Deconv2d, "use_k16", on)
        y = blk(x, skip)
        res[on] = (y.detach(),) + torch.autograd.grad(y, [x, skip, blk.conv.weight], cot)
    for name, a, b in zip(("value", "grad x", "grad skip", "grad weight"), res[True], res[False]):
        _close(name, a, b.cpu(), 2e-4 if name == "grad weight" else 5e-5)


@pytest.mark.parametrize("cp,cq,d,stride", [(16, 8, 128, 2), (32, 16, 64, 2), (64, 32, 32, 2), (8, 8, 256, 1), (4, 8, 256, 1), (4, 16, 128, 1)])
def test_matrix_core_weight_gradients_agree_with_the_vector_kernel_at_the_unet_shapes(cp, cq, d, stride, monkeypatch):
    """The U-Net's own layer shapes (P = coarse / output side, Q = fine / input side; volume_dims 256 / 128 / 64): the matrix-core weight-gradient kernels --
    stride 2 (conv3d_wgrad2_mfma_k), stride 1 (conv3d_wgrad_mfma_k) and its two-rows-per-wave form for the four-channel heads -- against the vector-ALU
    kernel on the same tensors, to float32 summation error of 2 M - 17 M terms per weight."""
    from gens_amd.ops.conv3d import _conv_wgrad
    g = torch.Generator(device="cuda").manual_seed(cp * 1000 + cq + d)
    p = torch.randn(cp, d, d, d, device="cuda", generator=g)
    q = torch.randn(cq, stride * d, stride * d, stride * d, device="cuda", generator=g)
    fast = _conv_wgrad(p, q, stride)
    monkeypatch.setenv("GENS_K15_NO_MFMA_WGRAD", "1")
    ref = _conv_wgrad(p, q, stride)
    assert fast.shape == ref.shape == (cp, cq, 27)
    err = float((fast - ref).abs().max() / ref.abs().max())
    assert err < 5e-6, err
