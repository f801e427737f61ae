"""K22 (gens_batchnorm2d_train_*): training-mode nn.BatchNorm2d [+ ReLU] of the MnasNet trunk against ATen's batch_norm in float64 on the CPU --
value, running statistics, the batch counter, and the gradients with respect to the input, the weight and the bias."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(5, 32, 24, 40), (2, 7, 5, 3), (1, 3, 1, 2), (5, 1152, 15, 20), (3, 16, 97, 131), (5, 48, 120, 160), (16, 4, 9, 9)]


def _pair(bn):
    """The norm with the activation module that follows it in the trunk's Sequential (feature_network._bn_relu)."""
    from gens_amd.models.modules.feature_network import ReLUAfterNorm
    return torch.nn.Sequential(bn, ReLUAfterNorm(bn) if bn.fused_relu else torch.nn.Identity())


@pytest.mark.parametrize("two_launches", [False, True])
@pytest.mark.parametrize("relu", [True, False])
def test_batchnorm_train_matches_aten_float64(relu, two_launches, monkeypatch):
    """(maps of at most 8 192 elements per channel take the one-launch kernels; GENS_K22_TWO_LAUNCHES sends them through the two-pass pair too)"""
    from gens_amd.models.modules.feature_network import BatchNorm2dReLU
    if two_launches:
        monkeypatch.setenv("GENS_K22_TWO_LAUNCHES", "1")
    for i, (n, c, h, w) in enumerate(SHAPES):
        g = torch.Generator().manual_seed(7 * i + relu)
        x = torch.randn(n, c, h, w, generator=g) * 2.0 + 0.5
        bn = BatchNorm2dReLU(c, relu=relu, momentum=0.01)
        with torch.no_grad():
            bn.weight.copy_(torch.randn(c, generator=g))
            bn.bias.copy_(torch.randn(c, generator=g) * 0.3)
            bn.running_mean.copy_(torch.randn(c, generator=g))
            bn.running_var.copy_(torch.rand(c, generator=g) + 0.5)
        ref_pair, dev_pair = _pair(copy.deepcopy(bn).double().train()), _pair(copy.deepcopy(bn).cuda().train())
        ref, dev = ref_pair[0], dev_pair[0]
        go = torch.randn(n, c, h, w, generator=g)
        xr = x.double().requires_grad_(True)
        yr = ref_pair(xr)                                    # CPU: nn.BatchNorm2d's own forward, then the activation module
        gr = torch.autograd.grad(yr, [xr, ref.weight, ref.bias], go.double())
        xd = x.cuda().requires_grad_(True)
        yd = dev_pair(xd)
        gd = torch.autograd.grad(yd, [xd, dev.weight, dev.bias], go.cuda())
        tag = (n, c, h, w)
        for name, a, b, tol in (("value", yd, yr, 2e-5), ("grad x", gd[0], gr[0], 1e-4), ("grad weight", gd[1], gr[1], 1e-4), ("grad bias", gd[2], gr[2], 1e-4),
                                ("running_mean", dev.running_mean, ref.running_mean, 1e-6), ("running_var", dev.running_var, ref.running_var, 1e-5)):
            err = float((a.detach().cpu().double() - b.detach()).abs().max())
            scale = max(1.0, float(b.abs().max()))
            assert err <= tol * scale, (name, tag, err, scale)
        assert int(dev.num_batches_tracked) == 1 and int(ref.num_batches_tracked) == 1
        # a second batch moves the running statistics again; eval mode then uses them (nn.BatchNorm2d's own path on both sides)
        dev_pair(x.cuda() * 0.5)
        ref_pair(x.double() * 0.5)
        assert int(dev.num_batches_tracked) == 2
        assert float((dev.running_var.cpu().double() - ref.running_var).abs().max()) <= 1e-5 * max(1.0, float(ref.running_var.abs().max()))
        dev.eval()
        ref.eval()
        assert float((dev_pair(x.cuda()).cpu().double() - ref_pair(x.double())).abs().max()) <= 2e-5 * max(1.0, float(ref_pair(x.double()).abs().max()))


def test_batchnorm_module_keeps_the_state_dict_and_declines_what_it_does_not_cover():
    from gens_amd import ops
    from gens_amd.models.modules.feature_network import BatchNorm2dReLU
    bn = BatchNorm2dReLU(6, relu=True, momentum=0.01)
    assert set(bn.state_dict()) == set(torch.nn.BatchNorm2d(6).state_dict())
    x = torch.randn(2, 6, 4, 4)
    assert not ops.batchnorm_supported(x, bn)                           # CPU tensor: nn.BatchNorm2d's forward
    assert ops.batchnorm_supported(x.cuda(), bn.cuda())
    bn.eval()
    assert not ops.batchnorm_supported(x.cuda(), bn)                    # eval mode: running statistics, ATen's kernel
    y = _pair(bn)(x.cuda())
    assert float(y.min()) >= 0.0
    nb = BatchNorm2dReLU(6, momentum=None).cuda().train()               # cumulative average: not covered
    assert not ops.batchnorm_supported(x.cuda(), nb)
