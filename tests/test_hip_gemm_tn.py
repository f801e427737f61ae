"""K14 (gens_gemm_tn): a^T b for tall operands against torch.matmul, and the Linear layer built on it against nn.Linear through three
orders of differentiation (what the SDF network's smooth term needs)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("k,m,n", [(61835, 128, 188), (247340, 32, 32), (247340, 33, 32), (20000, 101, 27), (9001, 1, 23), (8192, 8, 16), (70001, 64, 69),
                                   (12345, 257, 3)])
def test_gemm_tn_matches_torch(k, m, n):
    from gens_amd import ops
    g = torch.Generator(device="cuda").manual_seed(k + m + n)
    a = torch.randn(k, m, device="cuda", generator=g)
    b = torch.randn(k, n, device="cuda", generator=g)
    c = ops.matmul_tn(a, b)
    ref = (a.double().t() @ b.double())
    scale = float(ref.abs().max())
    assert c.shape == (m, n)
    assert float((c.double() - ref).abs().max()) <= 2e-6 * scale + 1e-4, float((c.double() - ref).abs().max())
    assert torch.equal(c, ops.matmul_tn(a, b))                         # partial sums are added in slab order: run-to-run identical


def test_linear_layer_gradients_match_nn_linear_to_third_order():
    from gens_amd.models.modules.linear import Linear
    torch.manual_seed(0)
    mine, ref = Linear(27, 101).cuda(), torch.nn.Linear(27, 101).cuda()
    ref.load_state_dict(mine.state_dict())
    head_m, head_r = Linear(101, 1).cuda(), torch.nn.Linear(101, 1).cuda()
    head_r.load_state_dict(head_m.state_dict())
    x = torch.randn(20000, 27, device="cuda")

    def run(l1, l2):
        xi = x.clone().requires_grad_(True)
        y = l2(torch.nn.functional.softplus(l1(xi), beta=10))
        g1 = torch.autograd.grad(y, xi, torch.ones_like(y), create_graph=True)[0]
        g2 = torch.autograd.grad(g1, xi, torch.ones_like(g1), create_graph=True)[0]
        loss = y.mean() + (g1.norm(dim=-1) - 1).pow(2).mean() + g2.norm(dim=-1).mean()
        params = list(l1.parameters()) + list(l2.parameters())
        return [y, g1, g2] + list(torch.autograd.grad(loss, params))

    for a, b in zip(run(mine, head_m), run(ref, head_r)):
        scale = float(b.abs().max()) + 1e-12
        assert float((a - b).abs().max()) <= 1e-4 * scale, (a.shape, float((a - b).abs().max()), scale)
