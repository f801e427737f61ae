"""Host logic of gens_sdf_value_f16 (no GPU): the weight stream and slot tables of gens_amd.ops._pack_value_units, executed by a
plain-torch model of the kernel's dataflow (k6v_sdf_value_f16.hip: transposed products, accumulator layout = next layer's B slots,
pre-scaled hidden units), must reproduce the network of /root/reference/models/modules/sdf_network.py:98-129."""
import math

import pytest
import torch

from gens_amd.ops import _pack_value_stream, _pack_value_units, _value_pairs, _value_slots


def network(ws, bs, pe, cond):
    sp = torch.nn.Softplus(beta=100)
    h = sp(pe @ ws[0].T + bs[0])
    for l in range(1, 6):
        if l == 3:
            h = torch.cat([h, pe], 1) / math.sqrt(2.0)
        h = sp(torch.cat([h, cond], 1) @ ws[l].T + bs[l])
    return torch.cat([h, cond], 1) @ ws[6][:1].T + bs[6][:1]


def kernel_model(stream, w_out, b_last, n_levels, pe, cond):
    """What the wave computes, in float64 on the de-quantised stream."""
    hid, pet, condt = _value_slots(n_levels)
    a = stream.double().sum(2)                                     # hi + lo: (U, 4, 64, 8)
    a = a.reshape(a.shape[0], 4, 2, 32, 8)                         # [unit][tile][half][m][slot]
    n = pe.shape[0]
    one, zero = torch.ones(n, 1, dtype=torch.float64), torch.zeros(n, 1, dtype=torch.float64)

    def operand(table, values):                                    # (blocks, 2, 8) slots of every point: (N, blocks, 2, 8)
        src = torch.cat([values, one, zero], 1)
        k = values.shape[1]
        cols = torch.where(table >= 0, table, torch.where(table == -1, torch.full_like(table, k), torch.full_like(table, k + 1)))
        return src[:, cols.reshape(-1)].reshape(n, *table.shape)

    def product(u0, b):                                            # -> (N, 128) pre-activations, feature 32 t + m
        blocks = b.shape[1]
        return torch.einsum("bthms,nbhs->ntm", a[u0:u0 + blocks], b).reshape(n, 128)

    def act(t):                                                    # c * softplus(t / c)
        return torch.where(t > 0.2 * 100 / math.log(2.0), t, torch.log2(1 + torch.exp2(t)))

    bp, bc = operand(pet, pe.double()), operand(condt, cond.double())
    nc = condt.shape[0]
    h = act(product(0, bp))
    u = 2
    for l in range(1, 6):
        t = product(u, operand(hid, h))
        u += 8
        if l == 3:
            t = t + product(u, bp)
            u += 2
        t = t + product(u, bc)
        u += nc
        h = act(t)
    assert u <= stream.shape[0] and stream.shape[0] % 4 == 0
    out = torch.zeros(n, dtype=torch.float64)
    for hh in range(2):
        feat = torch.tensor([32 * t + 8 * (r >> 2) + 4 * hh + (r & 3) for t in range(4) for r in range(16)])
        out += h[:, feat] @ w_out[hh, :64].double()
        tb = condt[:, hh].reshape(-1)
        out += (torch.where(tb >= 0, 1.0, 0.0) * cond.double()[:, tb.clamp(min=0)]) @ w_out[hh, 64:].double()
    return out + b_last


@pytest.mark.parametrize("n_levels", [3, 5])
def test_value_stream_reproduces_the_network(n_levels):
    g = torch.Generator().manual_seed(7 + n_levels)
    fe = 20 * n_levels
    dims = [(128, 27), (128, 128 + fe), (101, 128 + fe), (128, 128 + fe), (128, 128 + fe), (128, 128 + fe), (1 + 12, 128 + fe)]
    ws = [torch.randn(o, i, generator=g) / math.sqrt(i) for o, i in dims]
    bs = [0.1 * torch.randn(o, generator=g) for o, _ in dims]
    n = 37
    pe = torch.randn(n, 27, generator=g)
    cond = torch.randn(n, fe, generator=g)
    stream, w_out, vmax = _pack_value_units(ws, bs, n_levels)
    assert stream.dtype == torch.float16 and stream.shape[1:] == (4, 2, 64, 8) and vmax < 6e4
    want = network([w.double() for w in ws], [b.double() for b in bs], pe.double(), cond.double())[:, 0]
    got = kernel_model(stream, w_out, float(bs[6][0]), n_levels, pe, cond)
    # the stream is the weights rounded to (hi, lo) half pairs: ~2^-22 relative per weight
    assert (got - want).abs().max() < 2e-5 * want.abs().max().clamp(min=1.0)


def test_slot_tables_cover_every_column_once():
    for n_levels in (3, 5):
        hid, pe, cond = _value_slots(n_levels)
        assert sorted(hid.reshape(-1).tolist()) == list(range(128))
        assert sorted(v for v in pe.reshape(-1).tolist() if v >= 0) == list(range(27))
        assert sorted(v for v in cond.reshape(-1).tolist() if v >= 0) == list(range(20 * n_levels))
        assert (pe == -1).sum() == 1 and (cond == -1).sum() == 1


def pair_model(stream, w_out, b_last, n_levels, pe, cond):
    """k6t_sdf_value.hip's dataflow in float64: group g, tile T, lane (m, half), position i multiplies what lane half holds for pair i."""
    hid, pet, condt = _value_pairs(n_levels)
    a = stream.double().reshape(stream.shape[0], 4, 2, 32, 4)      # [group][tile][half][m][i]
    n = pe.shape[0]
    one, zero = torch.ones(n, 1, dtype=torch.float64), torch.zeros(n, 1, dtype=torch.float64)

    def operand(table, values):
        src = torch.cat([values, one, zero], 1)
        k = values.shape[1]
        cols = torch.where(table >= 0, table, torch.where(table == -1, torch.full_like(table, k), torch.full_like(table, k + 1)))
        return src[:, cols.reshape(-1)].reshape(n, *table.shape)

    def product(g0, b, issued=None):
        """issued[g] MFMAs of group g are executed (the kernel drops a block's trailing pair)."""
        groups = b.shape[1]
        w = a[g0:g0 + groups].clone()
        if issued is not None:
            for g, cnt in enumerate(issued):
                w[g, :, :, :, cnt:] = 0.0
        return torch.einsum("gthmi,nghi->ntm", w, b).reshape(n, 128)

    def act(t):
        return torch.where(t > 0.2 * 100 / math.log(2.0), t, torch.log2(1 + torch.exp2(t)))

    gc = condt.shape[0]
    ncs = 5 * 2 * n_levels + 1
    cond_issued = [4 if 4 * g + 4 <= ncs else 3 for g in range(gc)]
    bp, bc = operand(pet, pe.double()), operand(condt, cond.double())
    h = act(product(0, bp, [4, 4, 4, 3]))
    g = 4
    for l in range(1, 6):
        bh = operand(hid, h)
        if l == 3:
            t = product(g, bh[:, :13])
            g += 13
            t = t + product(g, bp, [4, 4, 4, 3])
            g += 4
        else:
            t = product(g, bh)
            g += 16
        t = t + product(g, bc, cond_issued)
        g += gc
        h = act(t)
    assert g + 1 == stream.shape[0] and float(stream[-1].abs().max()) == 0.0
    out = torch.zeros(n, dtype=torch.float64)
    for hh in range(2):
        feat = torch.tensor([32 * t + 8 * (r >> 2) + 4 * hh + (r & 3) for t in range(4) for r in range(16)])
        out += h[:, feat] @ w_out[hh, :64].double()
        tb = condt[:, hh].reshape(-1)
        out += (torch.where(tb >= 0, 1.0, 0.0) * cond.double()[:, tb.clamp(min=0)]) @ w_out[hh, 64:].double()
    return out + b_last


@pytest.mark.parametrize("n_levels", [1, 2, 3, 4, 5])
def test_float32_pair_stream_reproduces_the_network(n_levels):
    g = torch.Generator().manual_seed(70 + n_levels)
    fe = 20 * n_levels
    dims = [(128, 27), (128, 128 + fe), (101, 128 + fe), (128, 128 + fe), (128, 128 + fe), (128, 128 + fe), (13, 128 + fe)]
    ws = [torch.randn(o, i, generator=g) / math.sqrt(i) for o, i in dims]
    bs = [0.1 * torch.randn(o, generator=g) for o, _ in dims]
    n = 29
    pe = torch.randn(n, 27, generator=g)
    cond = torch.randn(n, fe, generator=g)
    stream, w_out = _pack_value_stream(ws, bs, n_levels)
    assert stream.dtype == torch.float32 and stream.shape[1:] == (4, 64, 4)
    want = network([w.double() for w in ws], [b.double() for b in bs], pe.double(), cond.double())[:, 0]
    got = pair_model(stream, w_out, float(bs[6][0]), n_levels, pe, cond)
    assert (got - want).abs().max() < 2e-6 * want.abs().max().clamp(min=1.0)      # float32 rounding of the scaled weights only


def test_pair_tables_cover_every_column_once():
    for n_levels in (1, 2, 3, 4, 5):
        hid, pe, cond = _value_pairs(n_levels)
        assert sorted(hid.reshape(-1).tolist()) == list(range(128))
        assert sorted(v for v in pe.reshape(-1).tolist() if v >= 0) == list(range(27))
        assert sorted(v for v in cond.reshape(-1).tolist() if v >= 0) == list(range(20 * n_levels))
        assert (pe == -1).sum() == 1 and (cond == -1).sum() == 1
        # the pair a kernel does not issue (trailing position of a block) carries nothing
        assert (pe[3, :, 3] == -2).all() and (cond[-1, :, 3] == -2).all()


def grad_model(stream, w_out, b_last, n_levels, pe, cond, dpe, dcond):
    """k6g_sdf_grad.hip's dataflow in float64 on the packed stream: forward chain (as pair_model, keeping softplus'), then the reverse
    chain G_{l-1} = (W_l^T G_l) * softplus' with the conditioning / point-encoding gradients accumulated in their slot-ordered tiles.
    dpe (N, 27, 3), dcond (N, FE, 3): derivatives of the network inputs with respect to x -> (sdf, d sdf / dx)."""
    hid, pet, condt = _value_pairs(n_levels)
    a = stream.double().reshape(stream.shape[0], 4, 2, 32, 4)      # [group][float4 / tile][half][m][i]
    n = pe.shape[0]
    one, zero = torch.ones(n, 1, dtype=torch.float64), torch.zeros(n, 1, dtype=torch.float64)
    c = 100.0 / math.log(2.0)
    nch = 2 * n_levels
    ncs = 5 * nch + 1
    gc_ = condt.shape[0]
    tc = ((5 * nch + 15) // 16 + 1) // 2 * 2                   # (in pairs: GradShapeT::TC)

    def operand(table, values):
        src = torch.cat([values, one, zero], 1)
        k = values.shape[1]
        cols = torch.where(table >= 0, table, torch.where(table == -1, torch.full_like(table, k), torch.full_like(table, k + 1)))
        return src[:, cols.reshape(-1)].reshape(n, *table.shape)

    def product(g0, b, issued=None):
        groups = b.shape[1]
        w = a[g0:g0 + groups].clone()
        if issued is not None:
            for g, cnt in enumerate(issued):
                w[g, :, :, :, cnt:] = 0.0
        return torch.einsum("gthmi,nghi->ntm", w, b).reshape(n, 128)

    def act(t):
        lin = t > 0.2 * c
        e = torch.exp2(torch.clamp(t, max=126.0))
        return torch.where(lin, t, torch.log2(1 + e)), torch.where(lin, torch.ones_like(t), e / (1 + e))

    cond_issued = [4 if 4 * g + 4 <= ncs else 3 for g in range(gc_)]
    bp, bc = operand(pet, pe.double()), operand(condt, cond.double())
    h, d0 = act(product(0, bp, [4, 4, 4, 3]))
    ds = [d0]
    g = 4
    for l in range(1, 6):
        bh = operand(hid, h)
        if l == 3:
            t = product(g, bh[:, :13])
            g += 13
            t = t + product(g, bp, [4, 4, 4, 3])
            g += 4
        else:
            t = product(g, bh)
            g += 16
        t = t + product(g, bc, cond_issued)
        g += gc_
        h, d = act(t)
        ds.append(d)
    sdf = torch.zeros(n, dtype=torch.float64)
    feat = [torch.tensor([32 * t_ + 8 * (r >> 2) + 4 * hh + (r & 3) for t_ in range(4) for r in range(16)]) for hh in range(2)]
    for hh in range(2):
        sdf += h[:, feat[hh]] @ w_out[hh, :64].double()
        tb = condt[:, hh].reshape(-1)
        sdf += (torch.where(tb >= 0, 1.0, 0.0) * cond.double()[:, tb.clamp(min=0)]) @ w_out[hh, 64:64 + tb.shape[0]].double()
    sdf = sdf + b_last

    # ---- reverse: G_5 = w_last * softplus'_5 in feature order
    w_hidden = torch.zeros(128, dtype=torch.float64)
    for hh in range(2):
        w_hidden[feat[hh]] = w_out[hh, :64].double() * c
    G = w_hidden * ds[5]
    # slot-ordered accumulators: gc[half][slot], gp[half][slot]; layer 6 contributes w_last directly
    gcond = torch.stack([w_out[hh, 64:64 + 16 * tc].double().expand(n, -1) for hh in range(2)], 1).clone()      # (N, 2, 16 tc)
    gpe = torch.zeros(n, 2, 16, dtype=torch.float64)

    def rows_to_slots(tile_vals):            # (N, 32) rows of an accumulator tile -> (N, 2 halves, 16 registers): m = 8 (r >> 2) + 4 half + (r & 3)
        out = torch.zeros(n, 2, 16, dtype=torch.float64)
        for hh in range(2):
            for r in range(16):
                out[:, hh, r] = tile_vals[:, 8 * (r >> 2) + 4 * hh + (r & 3)]
        return out

    for l in range(5, 0, -1):
        bg = operand(hid, G)                                         # G in the accumulator layout is the B operand
        ngroups = 13 if l == 2 else 16
        acc = product(g, bg[:, :ngroups])
        g += ngroups
        for cc in range(0, tc, 2):                                   # 2 tiles x 8 pairs per group: [tile cc: j = 0, 1; tile cc + 1: j = 0, 1]
            tiles = torch.zeros(n, 2, 32, dtype=torch.float64)
            t_list = [0, 1, 2] if l == 2 else [0, 1, 2, 3]
            specs = [(t_, gg, False) for t_ in t_list for gg in range(2)] + ([(3, 0, True)] if l == 2 else [])
            for t_, gg, half_only in specs:
                grp = a[g]                                           # (4 float4, 2 halves, 32 m, 4 i)
                g += 1
                for j in range(1 if half_only else 2):
                    bsel = bg[:, 4 * t_ + 2 * gg + j]                # (N, 2 halves, 4 i)
                    for k in range(2):
                        tiles[:, k] += torch.einsum("hmi,nhi->nm", grp[2 * k + j], bsel)
            for k in range(2):
                gcond[:, :, 16 * (cc + k):16 * (cc + k + 1)] += rows_to_slots(tiles[:, k])
        if l == 3:
            tile = torch.zeros(n, 32, dtype=torch.float64)
            for t_ in range(4):
                grp = a[g]
                g += 1
                for j in range(4):
                    tile += torch.einsum("hmi,nhi->nm", grp[j], bg[:, 4 * t_ + j])
            gpe += rows_to_slots(tile)
        G = acc * ds[l - 1]
    bg = operand(hid, G)
    tile = torch.zeros(n, 32, dtype=torch.float64)
    for t_ in range(4):
        grp = a[g]
        g += 1
        for j in range(4):
            tile += torch.einsum("hmi,nhi->nm", grp[j], bg[:, 4 * t_ + j])
    gpe += rows_to_slots(tile)
    assert g + 2 == stream.shape[0]
    # slots -> input columns, then the chain rule through the given input derivatives
    grad = torch.zeros(n, 3, dtype=torch.float64)
    cflat, pflat = condt.permute(1, 0, 2).reshape(2, -1), pet.permute(1, 0, 2).reshape(2, -1)
    for hh in range(2):
        for q in range(16 * tc):
            col = int(cflat[hh, q]) if q < cflat.shape[1] else -2
            if col >= 0:
                grad += gcond[:, hh, q, None] * dcond[:, col].double()
        for q in range(16):
            col = int(pflat[hh, q]) if q < pflat.shape[1] else -2
            if col >= 0:
                grad += gpe[:, hh, q, None] * dpe[:, col].double()
    return sdf, grad


@pytest.mark.parametrize("n_levels", [1, 2, 3, 4, 5])
def test_gradient_stream_reproduces_the_network_and_its_input_gradient(n_levels):
    from gens_amd.ops import _pack_grad_stream
    gen = torch.Generator().manual_seed(170 + n_levels)
    fe = 20 * n_levels
    dims = [(128, 27), (128, 128 + fe), (101, 128 + fe), (128, 128 + fe), (128, 128 + fe), (128, 128 + fe), (13, 128 + fe)]
    ws = [torch.randn(o, i, generator=gen) / math.sqrt(i) for o, i in dims]
    bs = [0.1 * torch.randn(o, generator=gen) for o, _ in dims]
    n = 11
    pe = torch.randn(n, 27, generator=gen).double().requires_grad_(True)
    cond = torch.randn(n, fe, generator=gen).double().requires_grad_(True)
    dpe = torch.randn(n, 27, 3, generator=gen)
    dcond = torch.randn(n, fe, 3, generator=gen)
    want = network([w.double() for w in ws], [b.double() for b in bs], pe, cond)[:, 0]
    gpe_, gcond_ = torch.autograd.grad(want.sum(), (pe, cond))
    want_grad = torch.einsum("nk,nkc->nc", gpe_, dpe.double()) + torch.einsum("nk,nkc->nc", gcond_, dcond.double())
    stream, w_out = _pack_grad_stream(ws, bs, n_levels)
    sdf, grad = grad_model(stream, w_out, float(bs[6][0]), n_levels, pe.detach(), cond.detach(), dpe, dcond)
    assert (sdf - want.detach()).abs().max() < 2e-6 * want.abs().max().clamp(min=1.0)
    assert (grad - want_grad).abs().max() < 5e-6 * want_grad.abs().max().clamp(min=1.0)
