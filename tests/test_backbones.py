"""The two CNNs around the hot path (gens_amd/models/modules/{reg_network,feature_network}.py) against the reference's classes
(golden g16, generated from reg_network.py:105-169 and feature_network_mnasnet.py:53-103 by tests/golden/make_golden.py g16):
same state-dict keys, same seeded initial weights, same outputs and gradients.  Plain PyTorch modules: CPU tests."""
import os

import numpy as np
import pytest
import torch

from gens_amd.config import Conf
from gens_amd.models import gens
from gens_amd.models.modules.feature_network import FeatureNetwork
from gens_amd.models.modules.reg_network import RegNetwork

from .conftest import GOLDEN


@pytest.fixture(scope="module")
def g16():
    d = np.load(os.path.join(GOLDEN, "g16_backbones.npz"))
    return {k: d[k] for k in d.files}


def test_reg_network_matches_the_reference_unet(g16):
    torch.manual_seed(160)
    net = RegNetwork(Conf({"d_voluem": [8, 8, 8], "d_out": [4, 4, 4], "d_base": 8})).eval()
    sd = net.state_dict()
    assert list(sd.keys()) == list(g16["reg.keys"])                       # a reference checkpoint loads with strict=True
    for k, v in sd.items():                                               # same construction order => same seeded initial weights
        assert torch.equal(v, torch.from_numpy(g16[f"reg.w.{k}"])), k
    vols = [torch.from_numpy(g16[f"reg.in{i}"]).requires_grad_(True) for i in range(3)]
    outs = net(vols)
    sum((o * torch.from_numpy(g16[f"reg.cot{i}"])).sum() for i, o in enumerate(outs)).backward()
    for i in range(3):
        assert outs[i].shape == (1, 4, 16 >> i, 16 >> i, 16 >> i)
        torch.testing.assert_close(outs[i], torch.from_numpy(g16[f"reg.out{i}"]), rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(vols[i].grad, torch.from_numpy(g16[f"reg.gin{i}"]), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(net.conv0.conv.weight.grad, torch.from_numpy(g16["reg.gw.conv0"]), rtol=1e-4, atol=1e-4)


def test_feature_network_matches_the_reference_wiring(g16):
    torch.manual_seed(162)
    net = FeatureNetwork(Conf({"d_out": [4, 4, 4, 4, 4]}))
    sd = net.state_dict()
    assert list(sd.keys()) == list(g16["feat.keys"])
    assert [str(tuple(v.shape)) for v in sd.values()] == list(g16["feat.shapes"])
    imgs = torch.from_numpy(g16["feat.imgs"])
    with torch.no_grad():
        for mode in ("eval", "train"):
            outs = net.train(mode == "train")(imgs)
            for i, o in enumerate(outs):
                assert o.shape == (2, 4, 64 >> i, 96 >> i)
                torch.testing.assert_close(o, torch.from_numpy(g16[f"feat.{mode}{i}"]), rtol=1e-4, atol=1e-5)
    # per-tensor checksums of the state after the two passes: the seeded weights, and the BatchNorm running statistics the train pass left
    sd = net.state_dict()
    np.testing.assert_allclose([float(v.double().sum()) for v in sd.values()], g16["feat.sums"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose([float(v.double().abs().sum()) for v in sd.values()], g16["feat.abs_sums"], rtol=1e-6, atol=1e-6)


def test_feature_network_trunk_is_mnasnet_1_0():
    """Parameter count and names of the published architecture (torchvision's MNASNet(1.0) has 4 383 312 parameters, of which the
    classifier 1 281 000 and the final 1x1 conv + BatchNorm 409 600 + 2 560 are not part of the reference's cut, :59-63)."""
    net = FeatureNetwork(Conf({"d_out": [4] * 5}))
    trunk = sum(p.numel() for n, p in net.named_parameters() if n.startswith("layer"))
    assert trunk == 4383312 - 1281000 - 409600 - 2560
    names = dict(net.named_parameters())
    assert names["layer1.0.weight"].shape == (32, 3, 3, 3) and names["layer1.6.weight"].shape == (16, 32, 1, 1)
    assert names["layer2.0.0.layers.0.weight"].shape == (48, 16, 1, 1) and names["layer2.0.0.layers.3.weight"].shape == (48, 1, 3, 3)
    assert names["layer3.0.2.layers.3.weight"].shape == (120, 1, 5, 5)
    assert names["layer4.1.1.layers.6.weight"].shape == (96, 576, 1, 1)
    assert names["layer5.0.3.layers.0.weight"].shape == (1152, 192, 1, 1) and names["layer5.1.0.layers.6.weight"].shape == (320, 1152, 1, 1)


def test_gens_builds_stand_alone_with_its_own_backbones():
    gens._BACKBONES.clear()
    from gens_amd.config import gens_model_conf
    model = gens.GenS(gens_model_conf(volume_dims=(16, 8, 4)))
    assert isinstance(model.feature_network, FeatureNetwork) and isinstance(model.reg_network, RegNetwork)
    assert isinstance(model.match_feature_network, FeatureNetwork) and not any(p.requires_grad for p in model.match_feature_network.parameters())
    groups = model.get_optim_params({"mlp_lr": 5e-4, "feat_lr": 1e-3})
    assert len(groups) == 2 and len(groups[1]["params"]) == len(list(model.feature_network.parameters())) + len(list(model.reg_network.parameters()))


def test_trunk_keeps_its_activations_when_a_utility_replaces_the_norm_modules():
    """The ReLU that K22 folds into the trunk's BatchNorm stays a real module in the Sequential's slot: replacing every `_BatchNorm` with a
    plain one (what nn.SyncBatchNorm.convert_sync_batchnorm, batch-norm folding or quantisation preparation do) must not drop an activation,
    and the untouched trunk must equal the plain-module trunk (no activation applied twice with a different result either)."""
    import copy

    import torch.nn as nn
    from gens_amd.models.modules import feature_network as fn
    torch.manual_seed(0)
    trunk = nn.Sequential(*fn._mnasnet_trunk()).train()
    plain = copy.deepcopy(trunk)

    def swap(mod):
        for name, child in mod.named_children():
            if isinstance(child, nn.modules.batchnorm._BatchNorm):
                new = nn.BatchNorm2d(child.num_features, eps=child.eps, momentum=child.momentum)
                new.load_state_dict(child.state_dict())
                setattr(mod, name, new)
            else:
                swap(child)
    swap(plain)
    assert not any(isinstance(m, fn.BatchNorm2dReLU) for m in plain.modules())
    assert sum(isinstance(m, nn.ReLU) for m in plain.modules()) == sum(isinstance(m, nn.ReLU) for m in trunk.modules()) > 30
    assert list(plain.state_dict()) == list(trunk.state_dict())
    x = torch.rand(2, 3, 64, 96)
    a, b = trunk(x), plain(x)
    assert torch.equal(a, b)
    # the activations are really there: without them the output differs
    bare = copy.deepcopy(plain)
    for m in bare.modules():
        if isinstance(m, nn.ReLU):
            m.forward = lambda t: t
    assert not torch.allclose(bare(x), b)
