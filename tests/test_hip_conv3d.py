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
    (8, 8, (4, 6, 64), 1, True), (12, 5, (3, 3, 128), 1, False), (4, 4, (2, 5, 192), 1, False)])      # z % 64 == 0: wgrad's LDS neighbour exchange
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


@pytest.mark.parametrize("cin,cout,dims", [(8, 8, (8, 8, 8)), (32, 16, (4, 4, 4)), (16, 8, (6, 10, 34)), (5, 3, (3, 2, 7)), (1, 9, (1, 1, 1))])
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


def test_reg_network_on_the_device_matches_the_cpu_module():
    from gens_amd.config import Conf
    from gens_amd.models.modules.reg_network import RegNetwork
    torch.manual_seed(11)
    net = RegNetwork(Conf({"d_voluem": [8, 8, 8], "d_out": [4, 4, 4], "d_base": 8}))
    g = torch.Generator().manual_seed(12)
    vols = [torch.randn(1, 8, d, d, d, generator=g).requires_grad_(True) for d in (32, 16, 8)]
    cots = [torch.randn(1, 4, d, d, d, generator=g) for d in (32, 16, 8)]
    outs = net(vols)
    sum((o * c).sum() for o, c in zip(outs, cots)).backward()
    import copy
    dnet = copy.deepcopy(net).cuda()
    dnet.zero_grad()
    dvols = [v.detach().cuda().requires_grad_(True) for v in vols]
    douts = dnet(dvols)
    sum((o * c.cuda()).sum() for o, c in zip(douts, cots)).backward()
    for i in range(3):
        _close(f"out{i}", douts[i], outs[i], 1e-4)
        _close(f"gin{i}", dvols[i].grad, vols[i].grad, 2e-4)
    for (name, p), q in zip(net.named_parameters(), dnet.parameters()):
        _close(name, q.grad, p.grad, 5e-4)


@pytest.mark.parametrize("shape", [(1, 8, 16, 16, 16), (1, 3, 5, 7, 9), (1, 32, 4, 4, 4), (1, 1, 1, 1, 3), (1, 8, 64, 64, 64)])
def test_instnorm_relu_matches_torch(shape):
    """K16 against torch's float64 instance_norm + relu on the CPU: value and gradient."""
    from gens_amd import ops
    g = torch.Generator().manual_seed(sum(shape))
    x = (torch.randn(shape, generator=g) * 1.7 + 0.4).requires_grad_(True)
    cot = torch.randn(shape, generator=g)
    y = torch.relu(F.instance_norm(x.double(), eps=1e-5))
    (gx,) = torch.autograd.grad(y, x, cot.double())
    xd = x.detach().cuda().requires_grad_(True)
    yd = ops.instnorm_relu(xd, 1e-5)
    _close("value", yd, y, 1e-5)
    # an element whose xhat is within rounding of 0 may take the other ReLU branch: compare away from the kink
    (gd,) = torch.autograd.grad(yd, xd, cot.cuda())
    xh = F.instance_norm(x.detach().double(), eps=1e-5)
    keep = xh.abs() > 1e-5
    err = ((gd.cpu().double() - gx) * keep).abs().max().item() / gx.abs().max().item()
    assert err < 2e-5, err


def test_instnorm_relu_statistics_survive_a_large_offset():
    """Sums are accumulated in float64: mean 1000, standard deviation 1 (E[x^2] - E[x]^2 in float32 would keep no digit)."""
    from gens_amd import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 2, 32, 32, 32, generator=g) + 1000.0
    y = torch.relu(F.instance_norm(x.double()))
    _close("value", ops.instnorm_relu(x.cuda()), y, 2e-4)                            # x itself carries 6e-5 of rounding at 1000
