"""Host logic of gens_blend_views4 (no GPU): the weight stream and tables of gens_amd.ops._pack_blend_t, decoded by the layout the kernel
assumes (k7t_blend.hip: A fragment of (M tile T, group g) holds for lane (m, qk), j = 0..3 the weight of output feature
16 T + 4 (m & 3) + (m >> 2) and input slot 16 g + 4 j + qk) and applied in the kernel's order, must reproduce BlendingNetwork.forward
(/root/reference/models/modules/blending_network.py:69-118)."""
import pytest
import torch

from gens_amd.models.modules.blending_network import BlendingNetwork
from gens_amd.ops import _pack_blend_t


class Stream:
    def __init__(self, stream):
        self.s, self.pos = stream.double(), 0

    def product(self, m_tiles, n_quads):
        """-> dense (16 m_tiles, 4 n_quads) matrix of the next product of the stream"""
        lane = torch.arange(64)
        m, qk = lane & 15, lane >> 4
        w = torch.zeros(16 * m_tiles, 4 * ((n_quads + 3) // 4) * 4, dtype=torch.float64)
        for t in range(m_tiles):
            rows = 16 * t + 4 * (m & 3) + (m >> 2)
            for g in range((n_quads + 3) // 4):
                frag = self.s[self.pos]
                self.pos += 1
                for j in range(4):
                    w[rows, 16 * g + 4 * j + qk] = frag[:, j]
        return w[:, :4 * n_quads]


def quad_bias(tab, entry, m_tiles):
    """accumulator-layout table -> dense bias vector: [q][4 T + i] = b[16 T + 4 i + q]"""
    b = torch.zeros(16 * m_tiles, dtype=torch.float64)
    for q in range(4):
        for t in range(m_tiles):
            for i in range(4):
                b[16 * t + 4 * i + q] = tab[entry, q, 4 * t + i]
    return b


def quad_row(tab, entry, n):
    return torch.stack([tab[entry, q, :] for q in range(4)], 1).reshape(-1)[:n].double()        # [kq][q] -> w[4 kq + q]


@pytest.mark.parametrize("n_levels", [1, 3, 5])
def test_blend_stream_reproduces_the_network(n_levels):
    torch.manual_seed(n_levels)
    f = 3 + 4 * n_levels
    xq = n_levels + 1
    net = BlendingNetwork(d_feature=4 * n_levels).double()
    with torch.no_grad():
        for p in net.parameters():
            p.add_(0.05 * torch.randn_like(p))
    g = lambda mod: (mod.weight.detach().float(), mod.bias.detach().float())  # noqa: E731
    layers = dict(rd1=g(net.ray_dir_fc[0]), rd2=g(net.ray_dir_fc[2]), b1=g(net.base_fc[0]), b2=g(net.base_fc[2]), v1=g(net.vis_fc[0]),
                  v2=g(net.vis_fc[2]), u1=g(net.vis_fc2[0]), u2=g(net.vis_fc2[2]), r1=g(net.rgb_fc[0]), r2=g(net.rgb_fc[2]), r3=g(net.rgb_fc[4]))
    stream, tab = _pack_blend_t(layers, f)
    tab = tab.double()
    n, s_views = 9, 4
    rgb_feat = torch.rand(n, s_views, f, dtype=torch.float64)
    ray_diff = torch.randn(n, s_views, 4, dtype=torch.float64) * 0.3
    mask = (torch.rand(n, s_views) > 0.2).double()
    with torch.no_grad():
        want = net(rgb_feat, ray_diff, mask)

    st = Stream(stream)
    elu = torch.nn.functional.elu
    one = torch.ones(n, s_views, 1, dtype=torch.float64)
    xt = (xq + 3) // 4
    d = elu(ray_diff @ st.product(1, 1).T + quad_bias(tab, 0, 1))                                    # ray_dir_fc.0
    x = torch.cat([rgb_feat, one], -1)                                                                # slot F = the one
    x = x + elu(d @ st.product(xt, 4).T + quad_bias(tab, 1, xt))[..., :4 * xq]
    e = torch.exp(net.s.detach().abs() * (ray_diff[..., 3:4] - 1))
    w = (e - e.min(dim=1, keepdim=True)[0]) * mask[..., None]
    w = w / (w.sum(dim=1, keepdim=True) + 1e-8)
    mean = (x * w).sum(dim=1, keepdim=True)
    var = (w * (x - mean) ** 2).sum(dim=1, keepdim=True)
    per_point = torch.cat([mean, var], -1) @ st.product(4, 2 * xq).T                                  # once per point
    h1 = elu(per_point + x @ st.product(4, xq).T)                                                     # + x's columns and the bias slot
    h = elu(h1 @ st.product(2, 16).T + quad_bias(tab, 2, 2))
    g1 = elu((h * w) @ st.product(2, 8).T + quad_bias(tab, 3, 2))
    vis = torch.sigmoid(elu(g1 @ quad_row(tab, 7, 32) + float(layers["v2"][1][32]))) * mask
    h = h + elu(g1 @ st.product(2, 8).T + quad_bias(tab, 4, 2))
    g2 = elu((h * vis[..., None]) @ st.product(2, 8).T + quad_bias(tab, 5, 2))
    vis2 = torch.sigmoid(g2 @ quad_row(tab, 8, 32) + float(layers["u2"][1][0])) * mask
    zero = torch.zeros_like(one)
    r_in = torch.cat([h, vis2[..., None], ray_diff[..., 0:3], ray_diff[..., 3:4], one, zero, zero], -1)      # quads 8 and 9
    c1 = elu(r_in @ st.product(1, 10).T)
    c2 = elu(c1 @ st.product(1, 4).T + quad_bias(tab, 6, 1))
    score = c2[..., :8] @ quad_row(tab, 9, 8) + float(layers["r3"][1][0])
    score = score.masked_fill(mask == 0, -1e9)
    got = (rgb_feat[..., :3] * torch.softmax(score, dim=1)[..., None]).sum(dim=1)
    assert st.pos + 2 == stream.shape[0] and float(stream[-2:].abs().max()) == 0.0
    assert (got - want).abs().max() < 2e-6            # the stream is the float32 rounding of the float64 test network
