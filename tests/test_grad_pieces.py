"""Host logic of the split-half value + gradient kernel (gens_sdf_grad_f16, k6gh_sdf_grad_f16.hip): the piece stream of
gens_amd.ops._pack_grad_pieces, consumed on the CPU by a lane-for-lane emulation of the kernel's dataflow (the MFMA operand layouts,
the order of the K blocks, the accumulator rows of the conditioning / point-encoding gradients, the lane-local chain rule), must give
the value and d sdf / dx of the network it was packed from (float64 autograd on the plain layers: sdf_network.py:98-154).  Runs without a
GPU; the kernel itself is compared with gens_sdf_grad and the oracle in test_hip_sdfmlp.py."""
import math

import pytest
import torch

from gens_amd import ops

F64 = torch.float64


def _network(n_levels, seed):
    g = torch.Generator().manual_seed(seed)
    fe = 20 * n_levels
    dims = [(128, 27), (128, 128 + fe), (101, 128 + fe), (128, 128 + fe), (128, 128 + fe), (128, 128 + fe), (1, 128 + fe)]
    ws = [torch.randn(o, i, generator=g) * (1.5 / math.sqrt(i)) for o, i in dims]
    bs = [torch.randn(o, generator=g) * 0.05 for o, _ in dims]
    return ws, bs


def _softplus100(a):
    return torch.nn.functional.softplus(a, beta=100.0, threshold=20.0)


def _embed(x, octaves):
    out = [x]
    for k in range(octaves):
        out += [torch.sin(2.0 ** k * x), torch.cos(2.0 ** k * x)]
    return torch.cat(out, -1)


def _reference(ws, bs, scale, x, f0, jac):
    """sdf and d sdf / dx at points x (N, 3) for volume features f0 + jac (x - x0) (the trilinear look-up linearised at the point)."""
    ws, bs = [w.to(F64) for w in ws], [b.to(F64) for b in bs]
    xg = x.clone().to(F64).requires_grad_(True)
    feat = f0.to(F64) + torch.einsum("nca,na->nc", jac.to(F64), xg - x.to(F64))
    pe, cond = _embed(xg * scale, 4), _embed(feat, 2)
    h = _softplus100(pe @ ws[0].t() + bs[0])
    for l in range(1, 6):
        inp = torch.cat([h, pe], -1) / math.sqrt(2.0) if l == 3 else h
        h = _softplus100(torch.cat([inp, cond], -1) @ ws[l].t() + bs[l])
    sdf = (torch.cat([h, cond], -1) @ ws[6].t() + bs[6])[:, :1] / scale
    grad, = torch.autograd.grad(sdf.sum(), xg)
    return sdf.detach(), grad


def _emulate(pieces, w_out, b_last, scale, g_scale, n_levels, x, f0, jac):
    """One wavefront of sdf_grad_h_k on 32 points, lane for lane, in float64 (operands = hi + lo)."""
    cf = 4 * n_levels
    nch, mid = cf // 2, n_levels // 2
    nc, tc = (5 * nch + 1 + 7) // 8, (5 * nch + 15) // 16
    c = 100.0 / math.log(2.0)
    A = pieces.to(F64).reshape(-1, 2, 64, 8).sum(1)                  # (block x tile, lane, slot): hi + lo
    lane = torch.arange(64)
    pt, half = lane & 31, lane >> 5

    def mfma(acc, a, b):
        """acc (64 lanes, 16 regs) += tile(A (64, 8)) x B (64, 8): lane (m, kh) of A holds row m, K = 8 kh + s; lane (n, kh) of B column n."""
        am = torch.cat([a[:32], a[32:]], 1)                          # (32 rows, 16 k)
        bm = torch.cat([b[:32], b[32:]], 1)                          # (32 points, 16 k)
        out = am @ bm.t()                                            # (row, point)
        r = torch.arange(16)
        rows = 8 * (r[None] >> 2) + 4 * half[:, None] + (r[None] & 3)
        return acc + out[rows, pt[:, None]]

    # prologue: the B-operand slots of every lane
    xs = x.to(F64)[pt] * scale                                       # (64, 3)
    q = torch.zeros(64, 16, dtype=F64)
    for a in range(3):
        v = xs[:, a]
        lo_h, hi_h = half == 0, half == 1
        q[lo_h, a] = v[lo_h]
        q[lo_h, 3 + a], q[lo_h, 6 + a] = torch.sin(v[lo_h]), torch.cos(v[lo_h])
        q[lo_h, 9 + a], q[lo_h, 12 + a] = torch.sin(2 * v[lo_h]), torch.cos(2 * v[lo_h])
        q[hi_h, a], q[hi_h, 3 + a] = torch.sin(4 * v[hi_h]), torch.cos(4 * v[hi_h])
        q[hi_h, 6 + a], q[hi_h, 9 + a] = torch.sin(8 * v[hi_h]), torch.cos(8 * v[hi_h])
    q[half == 0, 15] = 1.0
    P = [q[:, :8], q[:, 8:]]
    # this lane's channels: whole levels below (half 0) / above (half 1) the middle one, then two channels of the middle level
    chan = torch.zeros(64, nch, dtype=torch.long)
    for h in range(2):
        for j in range(mid):
            for cc in range(4):
                chan[half == h, 4 * j + cc] = 4 * (mid + 1 + j if h else j) + cc
        for cc in range(2):
            chan[half == h, 4 * mid + cc] = 4 * mid + 2 * h + cc
    f = f0.to(F64)[pt[:, None], chan]                                # (64, nch)
    JL = jac.to(F64)[pt[:, None], chan]                              # (64, nch, 3)
    e = torch.zeros(64, 8 * nc, dtype=F64)
    for j in range(nch):
        e[:, 5 * j] = f[:, j]
        e[:, 5 * j + 1], e[:, 5 * j + 2] = torch.sin(f[:, j]), torch.cos(f[:, j])
        e[:, 5 * j + 3], e[:, 5 * j + 4] = torch.sin(2 * f[:, j]), torch.cos(2 * f[:, j])
    e[half == 0, 5 * nch] = 1.0
    C = [e[:, 8 * k:8 * k + 8] for k in range(nc)]
    wo = w_out.to(F64)[half]                                         # (64, 64 + 16 tc)
    s_val = (e[:, :5 * nch] * wo[:, 64:64 + 5 * nch]).sum(1)

    cursor = [0]

    def segment(accs, cnt, nt, B):
        for i in range(cnt):
            for t in range(nt):
                accs[t] = mfma(accs[t], A[cursor[0] + t], B[i])
            cursor[0] += nt
        return accs

    def softplus(t):
        u = 1.0 + torch.exp2(t.clamp(max=126.0))
        return torch.maximum(t, torch.log2(u)), torch.exp2(t.clamp(max=126.0)) / u

    def blocks_of(tiles):                                            # tile t, register r -> K block 2 t + (r >> 3), slot r & 7
        return [tiles[b >> 1][:, 8 * (b & 1):8 * (b & 1) + 8] for b in range(8)]

    zero4 = lambda: [torch.zeros(64, 16, dtype=F64) for _ in range(4)]
    acc = segment(zero4(), 2, 4, P)
    D = []
    H = None
    for l in range(6):
        if l > 0:
            acc = segment(zero4(), nc, 4, C)
            if l == 3:
                acc = segment(acc, 2, 4, P)
            acc = segment(acc, 8, 4, H)
        hd = [softplus(a) for a in acc]
        D.append([d for _, d in hd])
        if l < 5:
            H = blocks_of([h for h, _ in hd])
        else:
            for t in range(4):
                s_val = s_val + (hd[t][0] * wo[:, 16 * t:16 * t + 16]).sum(1)
            H = blocks_of([wo[:, 16 * t:16 * t + 16] * (c * g_scale) * hd[t][1] for t in range(4)])
    s_val = s_val[:32] + s_val[32:]
    sdf = (s_val + b_last) / scale

    gc = [torch.where((16 * k + torch.arange(16))[None] < 5 * nch, wo[:, 64 + 16 * k:64 + 16 * k + 16] * g_scale, torch.zeros(64, 16, dtype=F64)) for k in range(tc)]
    gp = torch.zeros(64, 16, dtype=F64)
    for l in range(5, 0, -1):
        nt = 4 + tc + (1 if l == 3 else 0)
        accs = zero4() + gc + ([gp] if l == 3 else [])
        accs = segment(accs, 7 if l == 2 else 8, nt, H)
        gc = accs[4:4 + tc]
        if l == 3:
            gp = accs[4 + tc]
        H = blocks_of([accs[t] * D[l - 1][t] for t in range(4)])
    gp = segment([gp], 8, 1, H)[0]
    assert cursor[0] * 2 <= pieces.shape[0] < cursor[0] * 2 + 8       # every piece consumed, padding only behind

    g = torch.zeros(64, 3, dtype=F64)
    for a in range(3):
        v = xs[:, a]
        g0 = gp[:, a] + (gp[:, 3 + a] * torch.cos(v) - gp[:, 6 + a] * torch.sin(v)) + 2.0 * (gp[:, 9 + a] * torch.cos(2 * v) - gp[:, 12 + a] * torch.sin(2 * v))
        g1 = 4.0 * (gp[:, a] * torch.cos(4 * v) - gp[:, 3 + a] * torch.sin(4 * v)) + 8.0 * (gp[:, 6 + a] * torch.cos(8 * v) - gp[:, 9 + a] * torch.sin(8 * v))
        g[:, a] = torch.where(half == 0, g0, g1) * scale
    gcat = torch.cat(gc, 1)
    for j in range(nch):
        k = 5 * j
        df = (gcat[:, k] + gcat[:, k + 1] * torch.cos(f[:, j]) - gcat[:, k + 2] * torch.sin(f[:, j])
              + 2.0 * (gcat[:, k + 3] * torch.cos(2 * f[:, j]) - gcat[:, k + 4] * torch.sin(2 * f[:, j])))
        g = g + df[:, None] * JL[:, j]
    g = (g[:32] + g[32:]) / (scale * g_scale)
    return sdf[:, None], g


@pytest.mark.parametrize("n_levels,seed", [(3, 0), (3, 1), (5, 2)])
def test_piece_stream_reproduces_the_network(n_levels, seed):
    ws, bs = _network(n_levels, seed)
    pieces, vmax = ops._pack_grad_pieces(ws, bs, n_levels)
    assert pieces.dtype == torch.float16 and pieces.shape[1:] == (64, 8) and pieces.shape[0] % 8 == 0 and vmax < 6.0e4
    _, w_out = ops._pack_grad_stream(ws, bs, n_levels)
    g = torch.Generator().manual_seed(100 + seed)
    x = torch.rand(32, 3, generator=g) * 1.6 - 0.8
    f0 = torch.randn(32, 4 * n_levels, generator=g) * 0.7
    jac = torch.randn(32, 4 * n_levels, 3, generator=g)
    scale = 2.5
    sdf_ref, grad_ref = _reference(ws, bs, scale, x, f0, jac)
    sdf, grad = _emulate(pieces, w_out, float(bs[6][0]), scale, 512.0, n_levels, x, f0, jac)
    # the stream holds float32 weights as hi + lo halfs: 2^-22 relative per weight
    assert (sdf - sdf_ref).abs().max() <= 2e-5 * sdf_ref.abs().max()
    assert (grad - grad_ref).abs().max() <= 2e-5 * grad_ref.abs().max()
