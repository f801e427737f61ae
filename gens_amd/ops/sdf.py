"""K6 (the fused SDF network in inference) and K17 (the SDF network of a training step).

Part of gens_amd.ops (see ops/__init__.py); citations are relative to /root/reference."""
from .base import *  # noqa: F401,F403

# ------------------------------------------------------------------------------------------------------------------
# K6  fused SDF network (inference): look-up + encodings + 7 layers on fp32 MFMA (+ d sdf/dx)   (sdf_network.py:98-146)
# ------------------------------------------------------------------------------------------------------------------
def _pack_b_fragments(w):
    """(J, K) matrix -> MFMA 32x32x2 B fragments [ceil(J/32)][ceil(K/2)][64]: lane l of fragment (nt, kk) holds
    w[32 nt + (l & 31)][2 kk + (l >> 5)] (zero padded), so one B operand is one contiguous 256-B load."""
    j, k = w.shape
    nt, kk = (j + 31) // 32, (k + 1) // 2
    wp = torch.zeros(nt * 32, kk * 2, device=w.device, dtype=_f32)
    wp[:j, :k] = w
    return wp.view(nt, 32, kk, 2).permute(0, 2, 3, 1).contiguous()


def _pack_b_groups(w):
    """(J, K) matrix -> grouped fp32 MFMA B stream for gens_sdf_mlp: [ceil(J/32)][ceil(K/8)][64][4]; lane l of group
    (nt, g) holds w[32 nt + (l & 31)][8 g + 4 (l >> 5) + 0..3] (zero padded): one global_load_dwordx4 feeds 4 MFMAs."""
    j, k = w.shape
    nt, g = (j + 31) // 32, (k + 7) // 8
    wp = torch.zeros(nt * 32, g * 8, device=w.device, dtype=_f32)
    wp[:j, :k] = w
    return wp.view(nt, 32, g, 2, 4).permute(0, 2, 3, 1, 4).contiguous()


def _pack_b16(w, groups):
    """(J <= 16, K) matrix -> B stream of a narrow layer for two 16x16x4 fp32 MFMA tiles (k7_blend.hip::narrow_group): for every group
    (k0, S) of 4 S reduction columns, 64 lanes x S floats; lane l holds w[l % 16][k0 + S (l // 16) + 0..S-1] (zero padded)."""
    j, k = w.shape
    assert j <= 16
    kmax = max(k0 + 4 * s for k0, s in groups)
    wp = torch.zeros(16, kmax, device=w.device, dtype=_f32)
    wp[:j, :k] = w
    parts = []
    for k0, s in groups:
        blk = wp[:, k0:k0 + 4 * s].reshape(16, 4, s)           # [j][q][s]
        parts.append(blk.permute(1, 0, 2).reshape(-1))         # lane = q * 16 + j
    return torch.cat(parts).contiguous()


def _value_slots(n_levels):
    """Which input column every B-operand slot of k6v_sdf_value_f16.hip carries: three tables of shape (blocks, half, 8) holding a column
    number, -1 for the constant-one slot and -2 for a zero slot.  Hidden blocks: the accumulator layout of the previous layer (lane half h,
    register r of tile t = feature 32 t + 8 (r >> 2) + 4 h + (r & 3) = slot r & 7 of block 2 t + (r >> 3)).  Point encoding: half 0 holds
    pe[0:15] and the one, half 1 pe[15:27].  Volume features: half 0 the channels of the levels below the middle one and its first two, half
    1 the levels above and its last two; five encodings per channel (column e * CF + channel, sdf_network.py:104-107), then the one."""
    cf = 4 * n_levels
    nch, mid = cf // 2, n_levels // 2
    nc = (5 * nch + 1 + 7) // 8
    hid = torch.tensor([[[32 * (b >> 1) + 16 * (b & 1) + 8 * (s >> 2) + 4 * h + (s & 3) for s in range(8)] for h in range(2)] for b in range(8)])
    pe = torch.full((2, 2, 8), -2, dtype=torch.long)
    for q in range(16):
        pe[q >> 3, 0, q & 7] = q if q < 15 else -1
        if q < 12:
            pe[q >> 3, 1, q & 7] = 15 + q
    cond = torch.full((nc, 2, 8), -2, dtype=torch.long)
    odd = n_levels % 2
    for h in range(2):
        nfull = 4 * mid                                       # whole levels of a half: the first / the last n_levels // 2
        for lc in range(nch):
            ch = (lc if h == 0 else 4 * (mid + odd) + lc) if lc < nfull else 4 * mid + 2 * h + (lc - nfull)
            for e in range(5):
                q = 5 * lc + e
                cond[q >> 3, h, q & 7] = e * cf + ch
    cond[(5 * nch) >> 3, 0, (5 * nch) & 7] = -1
    return hid, pe, cond


def _pack_value_units(ws, bs, n_levels):
    """The weight stream and the output row of gens_sdf_value_f16 (layout and scaling: k6v_sdf_value_f16.hip's header).  ws[l] (out_l, in_l)
    and bs[l] are the effective float32 weights of lin0..lin6.  Returns (units (U, 4, 2, 64, 8) float16, w_out (2, 64 + 8 NC) float32,
    largest magnitude handed to half precision)."""
    dev = ws[0].device
    c = 100.0 / math.log(2.0)
    r2 = 1.0 / math.sqrt(2.0)
    hid, pe, cond = (t.to(dev) for t in _value_slots(n_levels))
    fe = 20 * n_levels

    def block_units(aug, table, offset):
        """aug: (128, K + 2) with the bias in column K and zeros in column K + 1; table entries index aug[:, offset + entry]."""
        k = aug.shape[1] - 2
        cols = torch.where(table >= 0, table + offset, torch.where(table == -1, torch.full_like(table, k), torch.full_like(table, k + 1)))
        g = aug[:, cols.reshape(-1)].reshape(4, 32, *table.shape)               # [tile][m][block][half][slot]
        return g.permute(2, 0, 3, 1, 4).reshape(table.shape[0], 4, 64, 8)        # [block][tile][lane = 32 half + m][slot]

    units = []
    zero = torch.zeros(128, 1, device=dev, dtype=_f32)
    for l in range(6):
        w = torch.zeros(128, ws[l].shape[1], device=dev, dtype=_f32)
        w[:ws[l].shape[0]] = ws[l]
        b = torch.zeros(128, 1, device=dev, dtype=_f32)
        b[:bs[l].shape[0], 0] = bs[l]
        if l == 0:
            units.append(block_units(torch.cat([c * w, c * b, zero], 1), pe, 0))
            continue
        h = w[:, :128].clone()
        if l == 3:                                                               # x = cat([h[:101], pe]) / sqrt(2)   (sdf_network.py:111-112)
            skip = torch.cat([c * r2 * w[:, 101:128], zero, zero], 1)            # the one slot of the point encoding carries nothing here
            h = r2 * h
            h[:, 101:] = 0.0
        aug = torch.cat([h, c * w[:, 128:], c * b, zero], 1)
        units.append(block_units(aug, hid, 0))
        if l == 3:
            units.append(block_units(skip, torch.where(pe == -1, torch.full_like(pe, -2), pe), 0))
        units.append(block_units(aug, cond, 128))
    units = torch.cat(units, 0)
    pad = (-units.shape[0]) % 4                                                  # whole chunks of four units
    if pad:
        units = torch.cat([units, torch.zeros(pad, *units.shape[1:], device=dev, dtype=_f32)], 0)
    hi = units.half()
    lo = (units - hi.float()).half()
    stream = torch.stack([hi, lo], 2).contiguous()                               # [unit][tile][hi, lo][lane][slot]
    w_last = ws[6][0]
    nc = cond.shape[0]
    w_out = torch.zeros(2, 64 + 8 * nc, device=dev, dtype=_f32)
    for hh in range(2):
        feat = torch.tensor([32 * t + 8 * (r >> 2) + 4 * hh + (r & 3) for t in range(4) for r in range(16)], device=dev)
        w_out[hh, :64] = w_last[feat] / c
        tb = cond[:, hh].reshape(-1)
        w_out[hh, 64:] = torch.where(tb >= 0, w_last[(128 + tb).clamp(0, 127 + fe)], torch.zeros_like(tb, dtype=_f32))
    return stream, w_out, float(units.abs().max())


def _value_pairs(n_levels):
    """Slot tables of k6t_sdf_value.hip, (groups, half, 4) each (column number, -1 = the constant one, -2 = zero): an MFMA of group g,
    position i multiplies the weights of the two columns [g, 0, i] and [g, 1, i] with what the two lane halves hold.  Hidden groups
    (t, g'): columns 32 t + 8 g' + 4 half + i (the accumulator layout).  Point encoding: half 0 pe[0:15], half 1 pe[15:27] and the one.
    Volume features: as gens_amd.ops._value_slots, 5 encodings per channel of the half, then the one (half 0)."""
    cf = 4 * n_levels
    nch, mid = cf // 2, n_levels // 2
    gc = (5 * nch + 1 + 3) // 4
    hid = torch.tensor([[[32 * t + 8 * g + 4 * h + i for i in range(4)] for h in range(2)] for t in range(4) for g in range(4)])
    pe = torch.full((4, 2, 4), -2, dtype=torch.long)
    for q in range(15):
        pe[q >> 2, 0, q & 3] = q
        pe[q >> 2, 1, q & 3] = 15 + q if q < 12 else (-1 if q == 12 else -2)
    cond = torch.full((gc, 2, 4), -2, dtype=torch.long)
    odd = n_levels % 2                                        # an odd level count: the middle level is shared, two channels per lane half
    for h in range(2):
        nfull = 4 * mid                                       # whole levels of a half: the first / the last n_levels // 2 (k6t_sdf_value.hip)
        for lc in range(nch):
            ch = (lc if h == 0 else 4 * (mid + odd) + lc) if lc < nfull else 4 * mid + 2 * h + (lc - nfull)
            for e in range(5):
                q = 5 * lc + e
                cond[q >> 2, h, q & 3] = e * cf + ch
    cond[(5 * nch) >> 2, 0, (5 * nch) & 3] = -1
    return hid, pe, cond


def _pack_value_stream(ws, bs, n_levels):
    """The float32 weight stream and output row of gens_sdf_value (k6t_sdf_value.hip): per group of four feature pairs and output tile
    T one float4 per lane (m, half) = the weights of row 32 T + m for the group's four columns of that half; columns fed by unscaled
    inputs carry 100 / ln 2 (pre-scaled hidden units), layer 3's hidden columns 1 / sqrt(2), its skip columns both; one zero group is
    appended because the kernel requests the next group before it knows there is none.  -> (stream (NG + 1, 4, 64, 4), w_out)."""
    dev = ws[0].device
    c = 100.0 / math.log(2.0)
    r2 = 1.0 / math.sqrt(2.0)
    hid, pe, cond = (t.to(dev) for t in _value_pairs(n_levels))
    fe = 20 * n_levels

    def groups(aug, table, offset):
        k = aug.shape[1] - 2
        cols = torch.where(table >= 0, table + offset, torch.where(table == -1, torch.full_like(table, k), torch.full_like(table, k + 1)))
        g = aug[:, cols.reshape(-1)].reshape(4, 32, *table.shape)               # [tile][m][group][half][i]
        return g.permute(2, 0, 3, 1, 4).reshape(table.shape[0], 4, 64, 4)        # [group][tile][lane = 32 half + m][i]

    out = []
    zero = torch.zeros(128, 1, device=dev, dtype=_f32)
    for l in range(6):
        w = torch.zeros(128, ws[l].shape[1], device=dev, dtype=_f32)
        w[:ws[l].shape[0]] = ws[l]
        b = torch.zeros(128, 1, device=dev, dtype=_f32)
        b[:bs[l].shape[0], 0] = bs[l]
        if l == 0:
            out.append(groups(torch.cat([c * w, c * b, zero], 1), pe, 0))
            continue
        h = w[:, :128].clone()
        if l == 3:                                                               # x = cat([h[:101], pe]) / sqrt(2)   (sdf_network.py:111-112)
            skip = torch.cat([c * r2 * w[:, 101:128], zero, zero], 1)
            h = r2 * h
            h[:, 101:] = 0.0
        aug = torch.cat([h, c * w[:, 128:], c * b, zero], 1)
        out.append(groups(aug, hid if l != 3 else hid[:13], 0))                  # layer 3 reads features 0..103 only
        if l == 3:
            out.append(groups(skip, torch.where(pe == -1, torch.full_like(pe, -2), pe), 0))
        out.append(groups(aug, cond, 128))
    out.append(torch.zeros(1, 4, 64, 4, device=dev, dtype=_f32))
    stream = torch.cat(out, 0).contiguous()
    w_last = ws[6][0]
    w_out = torch.zeros(2, 64 + 4 * cond.shape[0], device=dev, dtype=_f32)
    for hh in range(2):
        feat = torch.tensor([32 * t + 8 * (r >> 2) + 4 * hh + (r & 3) for t in range(4) for r in range(16)], device=dev)
        w_out[hh, :64] = w_last[feat] / c
        tb = cond[:, hh].reshape(-1)
        w_out[hh, 64:] = torch.where(tb >= 0, w_last[(128 + tb).clamp(0, 127 + fe)], torch.zeros_like(tb, dtype=_f32))
    return stream, w_out


def _pack_grad_stream(ws, bs, n_levels):
    """The weight stream and output row of gens_sdf_grad (k6g_sdf_grad.hip): the forward groups of _pack_value_stream, then the reverse
    pass on the TRUE (unscaled) transposed matrices, layer 5 down to 1: 16 groups (layer 2: 13) of W_l[:, :128]^T for the hidden-unit
    gradients, then per pair of conditioning tiles 8 groups (layer 2: 7) of 2 tiles x 8 pairs whose ROWS are ordered so that lane half h,
    register r of tile c receives the gradient of that half's slot 16 c + r, at layer 3 four groups of 1 tile x 16 pairs for the
    point-encoding slots, and after layer 1 the same four groups of W_0^T; two trailing zero groups (the kernel reads two groups ahead)."""
    dev = ws[0].device
    r2 = 1.0 / math.sqrt(2.0)
    c = 100.0 / math.log(2.0)
    hid, pe, cond = (t.to(dev) for t in _value_pairs(n_levels))
    nch = 2 * n_levels
    tc = ((5 * nch + 15) // 16 + 1) // 2 * 2                 # conditioning-gradient tiles, in pairs (k6g_sdf_grad.hip: GradShapeT::TC)
    fwd, _ = _pack_value_stream(ws, bs, n_levels)
    out = [fwd[:-1]]

    def groups(mat, table):
        """mat (32 NT, 128): rows = output rows of NT tiles, columns = hidden units of the layer -> (G, NT, 64, 4)."""
        nt = mat.shape[0] // 32
        g = mat[:, table.reshape(-1)].reshape(nt, 32, *table.shape)
        return g.permute(2, 0, 3, 1, 4).reshape(table.shape[0], nt, 64, 4)

    # accumulator row m of a tile <-> (lane half, register): m = 8 (r >> 2) + 4 half + (r & 3)
    m = torch.arange(32, device=dev)
    row_half, row_reg = (m >> 2) & 1, ((m >> 3) << 2) | (m & 3)
    cond_flat = cond.permute(1, 0, 2).reshape(2, -1)                     # [half][slot] -> feature column, -1 one, -2 nothing
    pe_flat = pe.permute(1, 0, 2).reshape(2, -1)
    for l in range(5, 0, -1):
        w = torch.zeros(128, ws[l].shape[1], device=dev, dtype=_f32)
        w[:ws[l].shape[0]] = ws[l]
        wt = w[:, :128].t().clone()                                       # rows: hidden inputs, columns: units of layer l
        if l == 3:
            wt = r2 * wt
            wt[101:] = 0.0
        out.append(groups(wt, hid if l != 2 else hid[:13]).reshape(-1, 4, 64, 4))
        mc = torch.zeros(32 * tc, 128, device=dev, dtype=_f32)
        for cc in range(tc):
            slot = 16 * cc + row_reg
            col = torch.where(slot < cond_flat.shape[1], cond_flat[row_half, slot.clamp(max=cond_flat.shape[1] - 1)], torch.full_like(slot, -2))
            live = col >= 0
            mc[32 * cc + m[live]] = w[:, 128 + col[live]].t()
        full = groups(mc, hid)                                            # (16 = (t, g), tc, 64, 4)
        for cc in range(0, tc, 2):
            for t in range(4 if l != 2 else 3):
                for gg in range(2):
                    a, b = full[4 * t + 2 * gg], full[4 * t + 2 * gg + 1]
                    out.append(torch.stack([a[cc], b[cc], a[cc + 1], b[cc + 1]])[None])
            if l == 2:
                a, b = full[12], full[13]
                out.append(torch.stack([a[cc], b[cc], a[cc + 1], b[cc + 1]])[None])
        if l == 3 or l == 1:
            src = r2 * w[:, 101:128] if l == 3 else None
            if l == 1:
                w0 = torch.zeros(128, 27, device=dev, dtype=_f32)
                w0[:ws[0].shape[0]] = ws[0]
                src = w0
            mp = torch.zeros(32, 128, device=dev, dtype=_f32)
            col = torch.where(row_reg < pe_flat.shape[1], pe_flat[row_half, row_reg.clamp(max=pe_flat.shape[1] - 1)], torch.full_like(row_reg, -2))
            live = col >= 0
            mp[m[live]] = src[:, col[live]].t()
            fp = groups(mp, hid)                                          # (16, 1, 64, 4)
            out.append(fp[:, 0].reshape(4, 4, 64, 4))                     # group t: the four float4 g = 0..3
    out.append(torch.zeros(2, 4, 64, 4, device=dev, dtype=_f32))
    stream = torch.cat(out, 0).contiguous()
    w_last = ws[6][0]
    fe = 20 * n_levels
    w_out = torch.zeros(2, 64 + 16 * tc, device=dev, dtype=_f32)
    for hh in range(2):
        feat = torch.tensor([32 * t + 8 * (r >> 2) + 4 * hh + (r & 3) for t in range(4) for r in range(16)], device=dev)
        w_out[hh, :64] = w_last[feat] / c
        tb = cond_flat[hh][:16 * tc]
        w_out[hh, 64:64 + tb.shape[0]] = torch.where(tb >= 0, w_last[(128 + tb).clamp(0, 127 + fe)], torch.zeros_like(tb, dtype=_f32))
    return stream, w_out


def _pack_grad_pieces(ws, bs, n_levels, n_pieces=None):
    """The piece stream of gens_sdf_grad_f16 (k6gh_sdf_grad_f16.hip): 1 KB pieces = the A operand (hi or lo halfs) of one 32-row output
    tile and one 16-deep K block, lane (m, kh) holding row m's weights for the eight reduction slots of lane half kh (_value_slots).
    Forward: layer 0's two point-encoding K blocks, then per layer the conditioning K blocks, (layer 3: the point-encoding blocks,) the
    eight hidden blocks -- the scaling of _pack_value_units, 4 tiles x {hi, lo} per block.  Reverse, on the TRUE transposed matrices,
    layer 5 down to 1: per K block of G_l (layer 2: seven) the four hidden tiles, the conditioning tiles and at layer 3 the
    point-encoding tile, rows ordered as in _pack_grad_stream; then the eight blocks of G_0 for the point-encoding tile.  Padded with
    zeros to whole chunks of eight pieces.  -> (pieces (N, 64, 8) float16, largest magnitude handed to half precision)."""
    dev = ws[0].device
    c = 100.0 / math.log(2.0)
    r2 = 1.0 / math.sqrt(2.0)
    hid, pe, cond = (t.to(dev) for t in _value_slots(n_levels))
    nch = 2 * n_levels
    tc = (5 * nch + 15) // 16
    zero = torch.zeros(128, 1, device=dev, dtype=_f32)

    def blocks(mat, table):
        """mat (32 NT, K + 2): column K = what the constant-one slot multiplies, column K + 1 zeros; table (B, 2, 8) of column numbers
        (-1 the one, -2 nothing) -> (B, NT, 64, 8): block, tile, lane = 32 half + m, slot."""
        nt, k = mat.shape[0] // 32, mat.shape[1] - 2
        cols = torch.where(table >= 0, table, torch.where(table == -1, torch.full_like(table, k), torch.full_like(table, k + 1)))
        g = mat[:, cols.reshape(-1)].reshape(nt, 32, *table.shape)                # [tile][m][block][half][slot]
        return g.permute(2, 0, 3, 1, 4).reshape(table.shape[0], nt, 64, 8)

    out = []
    for l in range(6):
        w = torch.zeros(128, ws[l].shape[1], device=dev, dtype=_f32)
        w[:ws[l].shape[0]] = ws[l]
        b = torch.zeros(128, 1, device=dev, dtype=_f32)
        b[:bs[l].shape[0], 0] = bs[l]
        if l == 0:
            out.append(blocks(torch.cat([c * w, c * b, zero], 1), pe).reshape(-1, 64, 8))
            continue
        h = w[:, :128].clone()
        if l == 3:                                                               # x = cat([h[:101], pe]) / sqrt(2)   (sdf_network.py:111-112)
            h = r2 * h
            h[:, 101:] = 0.0
        out.append(blocks(torch.cat([c * w[:, 128:], c * b, zero], 1), cond).reshape(-1, 64, 8))
        if l == 3:                                                               # (the one slot of the point encoding carries nothing here)
            out.append(blocks(torch.cat([c * r2 * w[:, 101:128], zero, zero], 1), torch.where(pe == -1, torch.full_like(pe, -2), pe)).reshape(-1, 64, 8))
        out.append(blocks(torch.cat([h, zero, zero], 1), hid).reshape(-1, 64, 8))
    fwd = torch.cat(out, 0)                                                       # (units x 4 tiles, 64, 8)
    # accumulator row m of a tile <-> (lane half, register): m = 8 (r >> 2) + 4 half + (r & 3)
    m = torch.arange(32, device=dev)
    row_half, row_reg = (m >> 2) & 1, ((m >> 3) << 2) | (m & 3)
    cond_flat = cond.permute(1, 0, 2).reshape(2, -1)                              # [half][slot] -> feature column, -1 one, -2 nothing
    pe_flat = pe.permute(1, 0, 2).reshape(2, -1)

    def slot_rows(src, flat, n_tiles):
        """(32 n_tiles, 128): row 32 c + m = src[:, column of slot 16 c + reg(m) of half(m)] (zero where the slot carries no column)."""
        mat = torch.zeros(32 * n_tiles, 128, device=dev, dtype=_f32)
        for cc in range(n_tiles):
            slot = 16 * cc + row_reg
            col = torch.where(slot < flat.shape[1], flat[row_half, slot.clamp(max=flat.shape[1] - 1)], torch.full_like(slot, -2))
            live = col >= 0
            mat[32 * cc + m[live]] = src[:, col[live]].t()
        return mat

    rev = []
    zz = torch.zeros(1, 2, device=dev, dtype=_f32)
    for l in range(5, 0, -1):
        w = torch.zeros(128, ws[l].shape[1], device=dev, dtype=_f32)
        w[:ws[l].shape[0]] = ws[l]
        wt = w[:, :128].t().clone()                                               # rows: hidden inputs, columns: units of layer l
        if l == 3:
            wt = r2 * wt
            wt[101:] = 0.0
        rows = [wt, slot_rows(w[:, 128:], cond_flat, tc)]
        if l == 3:
            rows.append(slot_rows(r2 * w[:, 101:128], pe_flat, 1))
        mat = torch.cat(rows, 0)
        mat = torch.cat([mat, zz.expand(mat.shape[0], 2)], 1)
        rev.append(blocks(mat, hid if l != 2 else hid[:7]).reshape(-1, 64, 8))  # [block][tile]
    w0 = torch.zeros(128, 27, device=dev, dtype=_f32)
    w0[:ws[0].shape[0]] = ws[0]
    mat = slot_rows(w0, pe_flat, 1)
    rev.append(blocks(torch.cat([mat, zz.expand(32, 2)], 1), hid).reshape(-1, 64, 8))
    tiles = torch.cat([fwd] + rev, 0)                                             # one row per (block, tile)
    hi = tiles.half()
    lo = (tiles - hi.float()).half()
    pieces = torch.stack([hi, lo], 1).reshape(-1, 64, 8)                          # [block][tile][hi, lo]
    pad = (-pieces.shape[0]) % 8 if n_pieces is None else n_pieces - pieces.shape[0]      # (whole chunks of the kernel's ring)
    if pad:
        pieces = torch.cat([pieces, torch.zeros(pad, 64, 8, device=dev, dtype=torch.float16)], 0)
    return pieces.contiguous(), float(tiles.abs().max())


class SdfMlpPlan:
    """Weights of an SDFNetwork re-packed for gens_sdf_mlp.  Only the shipped architecture is supported
    (`supported(net)`); anything else keeps using the PyTorch layers on top of the K2 look-up kernels."""

    @staticmethod
    def supported(net):
        return (net.num_layers == 8 and tuple(net.skip_in) == (3,) and net.embed_fn_fine is not None and net.embed_fn_feat is not None
                and net.lin0.weight_v.shape == (128, 27) and net.init_feat_channels in (4, 8, 12, 16, 20)
                and net.lin6.weight_v.shape[1] == 128 + 5 * net.init_feat_channels and net.lin2.weight_v.shape[0] == 101)

    @staticmethod
    def version(net):
        return tuple(p._version for p in net.parameters()) + tuple(p.data_ptr() for p in net.parameters())

    def __init__(self, net):
        assert SdfMlpPlan.supported(net), "gens_sdf_mlp is built for the architecture of confs/gens.conf:69-86"
        with torch.no_grad():
            ws, bs = [], []
            for l in range(7):
                lin = getattr(net, f"lin{l}")
                v, g = lin.weight_v.detach().to(_f32), lin.weight_g.detach().to(_f32)
                ws.append(v * (g / torch.linalg.norm(v, dim=1, keepdim=True)))
                bs.append(lin.bias.detach().to(_f32))
            dev = ws[0].device
            self.n_levels = net.init_feat_channels // 4
            self.wf, self.wb, self.bias = [], [], []
            c = 100.0 / math.log(2.0)      # pre-scaled forward streams (k6_sdfmlp.hip::softplus_t): hidden units travel as c * softplus
            for l in range(6):
                w = torch.zeros(128, ws[l].shape[1], device=dev, dtype=_f32)
                w[:ws[l].shape[0]] = ws[l]
                b = torch.zeros(128, device=dev, dtype=_f32)
                b[:bs[l].shape[0]] = bs[l]
                wbias = torch.cat([w, b[:, None]], 1)                              # bias = extra reduction row K_l (constant-1 input column)
                hidden = 0 if l == 0 else (101 if l == 3 else 128)                  # leading columns fed by (scaled) hidden units
                wbias[:, hidden:] *= c                                              # point encoding / skip columns / volume features / bias
                self.wf.append(_pack_b_groups(wbias))
                self.wb.append(_pack_b_groups(w.t().contiguous()))
                self.bias.append(b)
            # (the kernels' max / median activations drop NaNs: non-finite WEIGHTS are answered with NaN outputs, as the reference's layers would)
            self.finite = bool(torch.stack([torch.isfinite(w).all() for w in ws] + [torch.isfinite(b).all() for b in bs]).all())
            self.w_last = _c(ws[6][0].clone())
            self.w_last_scaled = self.w_last.clone()
            self.w_last_scaled[:128] /= c
            self.b_last = float(bs[6][0])
            self.scale = float(net.scale)
            self.value_stream, self.value_row = _pack_value_stream(ws, bs, self.n_levels)
            self.grad_stream, self.grad_row = _pack_grad_stream(ws, bs, self.n_levels)
            assert self.grad_stream.shape[0] == L.load().gens_sdf_grad_groups(self.n_levels) + 2
            # the split-half kernels (k6v / k6gh, the opt-in sdf_precision="f16x2") are built for 3 and 5 levels: other counts stay in float32
            self.value_units = self.value_w_out = None
            self.value_ok = False
            if self.n_levels in (3, 5):
                self.value_units, self.value_w_out, vmax = _pack_value_units(ws, bs, self.n_levels)
                self.value_ok = vmax < 6.0e4
            self.grad_pieces = None                        # the split-half value + gradient kernel: None if a weight leaves the half range
            n_pieces = L.load().gens_sdf_grad_f16_pieces(self.n_levels) if self.n_levels in (3, 5) else 0
            if n_pieces:
                self.grad_pieces, gvmax = _pack_grad_pieces(ws, bs, self.n_levels, n_pieces)
                assert self.grad_pieces.shape[0] == n_pieces
                top = float(ws[6][0, :128].abs().max())
                # gradients travel times a power of two that puts |w_last| near 256: lo parts of normal halfs, 128 x of head room
                self.grad_scale = 2.0 ** min(14, max(-10, round(math.log2(256.0 / top)))) if top > 0 and math.isfinite(top) else 1.0
                if not gvmax < 6.0e4:
                    self.grad_pieces = None
            self.overflow = torch.zeros(1, device=dev, dtype=torch.int32)
        self.wf_table, self.wb_table, self.bias_table = L.ptr_table(self.wf), L.ptr_table(self.wb), L.ptr_table(self.bias)
        self.key = SdfMlpPlan.version(net)

    def overflowed(self):
        """True if any split-half launch since the last call met a value outside the half range (synchronises)."""
        hit = bool(self.overflow.item())
        if hit:
            self.overflow.zero_()
        return hit


def sdf_mlp(plan, volumes, pts, index=None, want_grad=False, sdf_out=None, grad_out=None, precision="f32", count=None):
    """sdf (and d sdf/dx) of pts[index] written to sdf_out[index] / grad_out[index] (fresh, densely indexed outputs if
    no buffers are given).  volumes: packed VolumeSet with 3 or 5 levels.  No autograd graph is built (inference).
    precision: "f32" (exact float32 MFMA) or "f16x2" (split-half operands, ~1e-6 relative; check plan.overflowed()) -- value-only
    launches on gens_sdf_value_f16, value + gradient launches on gens_sdf_grad_f16 (float32 if the weights leave the half range).
    count: optional (1,) int32 device tensor from compact_valid(): only the first `count` entries of `index` are evaluated."""
    assert isinstance(volumes, VolumeSet) and volumes.layout == L.LAYOUT_PACKED and volumes.n == plan.n_levels
    pts = _c(pts.detach().reshape(-1, 3).to(_f32))
    n = pts.shape[0] if index is None else index.shape[0]
    if sdf_out is None:
        sdf_out = torch.empty(pts.shape[0], 1, device=pts.device, dtype=_f32)
    if want_grad and grad_out is None:
        grad_out = torch.empty(pts.shape[0], 3, device=pts.device, dtype=_f32)
    idx = None if index is None else _c(index.to(torch.int64))
    fe = 20 * plan.n_levels
    flops = 2 * (27 * 128 + (128 + fe) * (4 * 128 + 101 + 1)) * (2 if want_grad else 1)
    nbytes = n * (12 + (16 if want_grad else 4) + (8 if idx is not None else 0))
    tag = ":grad" if want_grad else ":value"     # profile key: the two device kernels (sdf_mlp_k<FE, true / false>) are priced separately
    if isinstance(plan, SdfTrainStep):           # this training step's streams (gens_sdf_train_pack): same layout, bias on the device
        L.call("gens_sdf_mlp_dev", volumes.table, volumes.dim_table, volumes.n, plan.wf_table, plan.wb_table, L.ptr(plan.w_last),
               L.ptr(plan.b_last), 1.0, L.ptr(pts), L.ptr(idx, torch.int64), n, L.ptr(count, torch.int32), L.ptr(sdf_out),
               L.ptr(grad_out) if want_grad else None, L.stream(), nbytes=nbytes, flops=n * flops, live=None if count is None else (count, n),
               label="gens_sdf_mlp" + tag)
        return (sdf_out, grad_out) if want_grad else sdf_out
    if want_grad and precision == "f16x2" and kernels.sdf_grad_f16 and getattr(plan, "grad_pieces", None) is not None:
        L.call("gens_sdf_grad_f16", volumes.table, volumes.dim_table, volumes.n, L.ptr(plan.grad_pieces, torch.float16), L.ptr(plan.grad_row), plan.b_last,
               plan.scale, plan.grad_scale, L.ptr(pts), L.ptr(idx, torch.int64), n, L.ptr(count, torch.int32), L.ptr(sdf_out), L.ptr(grad_out),
               L.ptr(sdf_grad_f16_stash(pts.device), torch.uint8), L.ptr(plan.overflow, torch.int32), L.stream(),
               nbytes=nbytes, flops=n * flops, live=None if count is None else (count, n), label="gens_sdf_grad_f16")
    elif want_grad and kernels.sdf_grad == "transposed":
        # (under "f16x2" with weights out of the half range this pass stays float32)
        L.call("gens_sdf_grad", volumes.table, volumes.dim_table, volumes.n, L.ptr(plan.grad_stream), L.ptr(plan.grad_row), plan.b_last,
               plan.scale, L.ptr(pts), L.ptr(idx, torch.int64), n, L.ptr(count, torch.int32), L.ptr(sdf_out), L.ptr(grad_out),
               L.ptr(sdf_grad_stash(pts.device), torch.uint8), L.stream(),
               nbytes=nbytes, flops=n * flops, live=None if count is None else (count, n), label="gens_sdf_grad")
    elif precision == "f16x2" and not want_grad and plan.value_ok:
        L.call("gens_sdf_value_f16", volumes.table, volumes.dim_table, volumes.n, L.ptr(plan.value_units, torch.float16), L.ptr(plan.value_w_out),
               plan.b_last, plan.scale, L.ptr(pts), L.ptr(idx, torch.int64), n, L.ptr(count, torch.int32), L.ptr(sdf_out),
               L.ptr(plan.overflow, torch.int32), L.stream(), nbytes=nbytes, flops=n * flops, live=None if count is None else (count, n),
               label="gens_sdf_value_f16")
    elif not want_grad and kernels.sdf_value == "transposed":
        L.call("gens_sdf_value", volumes.table, volumes.dim_table, volumes.n, L.ptr(plan.value_stream), L.ptr(plan.value_row), plan.b_last,
               plan.scale, L.ptr(pts), L.ptr(idx, torch.int64), n, L.ptr(count, torch.int32), L.ptr(sdf_out), L.stream(), nbytes=nbytes,
               flops=n * flops, live=None if count is None else (count, n), label="gens_sdf_value")
    else:
        L.call("gens_sdf_mlp", volumes.table, volumes.dim_table, volumes.n, plan.wf_table, plan.wb_table, L.ptr(plan.w_last),
               L.ptr(plan.w_last_scaled), plan.b_last, plan.scale, L.ptr(pts), L.ptr(idx, torch.int64), n, L.ptr(count, torch.int32), L.ptr(sdf_out),
               L.ptr(grad_out) if want_grad else None, L.stream(), nbytes=nbytes, flops=n * flops, live=None if count is None else (count, n),
               label="gens_sdf_mlp" + tag)
    if not getattr(plan, "finite", True):
        _poison(idx, count, sdf_out, grad_out if want_grad else None)
    return (sdf_out, grad_out) if want_grad else sdf_out


def _poison(idx, count, *outs):
    """NaN into the evaluated rows of `outs` (index map / device-side count as the kernels take them): a network with non-finite weights."""
    for out in outs:
        if out is None:
            continue
        if idx is None:
            out.fill_(float("nan"))
        else:
            live = idx if count is None else idx[torch.arange(idx.shape[0], device=idx.device) < count.to(idx.device)[0]]
            out[live] = float("nan")


# ------------------------------------------------------------------------------------------------------------------
# K17  the SDF network of a training step: value, gradient, `smooth`, and the loss backward   (sdf_network.py:98-154)
# ------------------------------------------------------------------------------------------------------------------
class SdfTrainStep:
    """One training / fine-tune step's view of the SDF network (gens_sdf_train_*): the effective (weight-normed) matrices are packed
    into MFMA B streams ONCE, then any number of point batches are evaluated against them.

        step = SdfTrainStep(weights, biases, volumes, packed)      # weights[l] (out_l, in_l) with autograd history, l = 0..6
        y, g, s = step(pts)                                         # (N,1), (N,3), (N,3); differentiable once more (loss.backward())
        g0 = step.first_order(pts0)                                 # d sdf / dx only, no graph (implicit_surface.py:305-310)

    `volumes`: the planar (1,4,X,Y,Z) tensors the gradient goes to; `packed`: their (X,Y,Z,4) texel copy the kernels read."""

    @staticmethod
    def supported(net, n_levels):
        return SdfMlpPlan.supported(net) and float(net.scale) == 1.0 and 1 <= n_levels <= 5 and net.init_feat_channels == 4 * n_levels

    def __init__(self, weights, biases, volumes, packed, raw=None, tv_masks=None):
        """weights / biases: the EFFECTIVE matrices with autograd history (torch._weight_norm's outputs) -- or, with raw = (weight_v list,
        weight_g list, bias list) of lin0..lin6, None: weight norm then happens inside the pack launch and its backward inside the
        gradient launch (gens_sdf_train_pack_wn / gens_sdf_train_wgrad), and the raw parameters are the autograd inputs.
        tv_masks: the mask pyramid; with it `step(pts, sel, tv=True)` also returns tv_regularization(volumes, masks) (implicit_surface.py:
        135-150) so that the dense TV gradient and the scattered look-up gradient of the volumes are formed in ONE buffer."""
        assert isinstance(packed, VolumeSet) and packed.layout == L.LAYOUT_PACKED and 1 <= packed.n <= 5
        self.volumes, self.packed = list(volumes), packed
        self.n_levels = packed.n
        self.raw = raw
        self.tv_masks = None if tv_masks is None else [_c(m.detach()) for m in tv_masks]
        dev = self.volumes[0].device if self.volumes else packed.tensors[0].device
        kin = 128 + 20 * self.n_levels
        self.kp = (kin + 1 + 7) // 8 * 8
        gf = [(27 + 1 + 7) // 8] + [self.kp // 8] * 5
        # backward n-tiles of layers 1..5: four hidden tiles + the conditioning tiles, padded to two or four (zero columns): the four waves of a
        # workgroup take one tile each or two share one (k17_sdf_train.hip: NT_B)
        n_cond = (20 * self.n_levels + 31) // 32
        ntb = [1] + [4 + (2 if n_cond <= 2 else 4)] * 5
        self.wf = [torch.empty(4 * g * 64 * 4, device=dev, dtype=_f32) for g in gf]
        self.wb = [torch.empty(nt * 16 * 64 * 4, device=dev, dtype=_f32) for nt in ntb]
        self.wf_table, self.wb_table = L.ptr_table(self.wf), L.ptr_table(self.wb)
        with torch.no_grad():
            if raw is None:
                assert len(weights) == 7 and len(biases) == 7
                self.tensors = [*weights, *biases]
                w = [_c(t.detach().to(_f32)) for t in weights[:6]]
                b = [_c(t.detach().to(_f32)) for t in biases[:6]]
                assert tuple(w[0].shape) == (128, 27) and tuple(w[2].shape) == (101, kin) and tuple(w[5].shape) == (128, kin)
                L.call("gens_sdf_train_pack", L.ptr_table(w), L.ptr_table(b), self.n_levels, self.wf_table, self.wb_table, L.stream())
                self.w_last = _c(weights[6].detach().to(_f32)[0].clone())
                self.b_last = _c(biases[6].detach().to(_f32)[:1].clone())
            else:
                vs, gs, bs = raw
                assert len(vs) == len(gs) == len(bs) == 7
                self.tensors = [*vs, *gs, *bs]
                self.v = [_c(t.detach().to(_f32)) for t in vs]
                self.g = [_c(t.detach().to(_f32).reshape(-1)) for t in gs]
                b = [_c(t.detach().to(_f32)) for t in bs]
                assert tuple(self.v[0].shape) == (128, 27) and tuple(self.v[2].shape) == (101, kin) and tuple(self.v[6].shape) == (129, kin)
                self.scale = [torch.empty(t.shape[0], device=dev, dtype=_f32) for t in self.v]
                self.w_last, self.b_last = torch.empty(kin, device=dev, dtype=_f32), torch.empty(1, device=dev, dtype=_f32)
                L.call("gens_sdf_train_pack_wn", L.ptr_table(self.v), L.ptr_table(self.g), L.ptr_table(b), self.n_levels, L.ptr_table(self.scale),
                       self.wf_table, self.wb_table, L.ptr(self.w_last), L.ptr(self.b_last), L.stream(), label="gens_sdf_train_pack")

    def _forward(self, pts, sel=None):
        """sel (StepPoints): evaluate pts[sel.idx[:count]] with the count left on the device and write rows sel.idx[i] of sel's dense
        outputs (their other rows already hold the reference's defaults); None: every row of pts, fresh outputs."""
        n = pts.shape[0]
        dev = pts.device
        stash = torch.empty(L.load().gens_sdf_train_stash_bytes(n, 0), device=dev, dtype=torch.uint8)
        if sel is None:
            y, g, s = (torch.empty(n, k, device=dev, dtype=_f32) for k in (1, 3, 3))
            idx = cnt = None
        else:
            y, g, s, idx, cnt = sel.y, sel.g, sel.s, sel.idx, sel.counts[0:1]
        fe = 20 * self.n_levels
        flops = 4 * 2 * (27 * 128 + (128 + fe) * (4 * 128 + 101 + 1))
        L.call("gens_sdf_train_fwd", self.packed.table, self.packed.dim_table, self.n_levels, self.wf_table, self.wb_table, L.ptr(self.w_last),
               L.ptr(self.b_last), L.ptr(pts), L.ptr(idx, torch.int64), n, L.ptr(cnt, torch.int32), L.ptr(stash, torch.uint8), L.ptr(y), L.ptr(g),
               L.ptr(s), L.stream(), nbytes=n * 40, flops=n * flops, live=None if cnt is None else (cnt, n))
        return y, g, s

    def __call__(self, pts, sel=None, tv=False):
        """-> (y, g, s) [, tv_reg when tv=True (needs tv_masks)]."""
        out = _SdfTrain.apply(_c(pts.detach().reshape(-1, 3).to(_f32)), self, sel, bool(tv), *self.tensors, *self.volumes)
        return out if tv else out[:3]

    @torch.no_grad()
    def first_order(self, pts):
        return self._forward(_c(pts.detach().reshape(-1, 3).to(_f32)))[1]


class _SdfTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pts, step, sel, tv, *tensors):
        # (NOT ctx.sel = sel: sel.y / sel.g / sel.s are this Function's OUTPUTS -- node -> sel -> output tensor -> grad_fn -> node is a cycle the
        # collector cannot break (the last edge lives in C++), so every training step would leave this node, the blending node behind sel.rgb and
        # everything they hold on the device for ever: 1.9 MB per step on the small test model, scripts/probe/eager_leak_probe.py)
        ctx.step = step
        ctx.sel_index = None if sel is None else (sel.idx, sel.counts[0:1])
        ctx.shapes = [t.shape for t in tensors]
        ctx.n_par = len(step.tensors)
        y, g, s = step._forward(pts, sel)
        tv_out = None
        if tv:
            assert step.tv_masks is not None, "SdfTrainStep(tv_masks=...) is needed for tv=True"
            nl = step.n_levels
            vols = [_c(v.detach()) for v in step.volumes]
            ctx.tv_dims = [d for v in vols for d in v.shape[-3:]]
            partial = torch.empty(L.load().gens_tv_levels_blocks(L.int_table(ctx.tv_dims), nl), 4, device=pts.device, dtype=_f32)
            tv_out = torch.empty(1 + nl, device=pts.device, dtype=_f32)
            L.call("gens_tv_levels_fwd", L.ptr_table(vols, align=16), L.ptr_table(step.tv_masks, align=16), L.int_table(ctx.tv_dims), nl, L.ptr(partial),
                   L.ptr(tv_out), L.stream(), nbytes=sum(20 * v[0, 0].numel() for v in vols))
            ctx.save_for_backward(pts, tv_out, *vols)
            return y, g, s, tv_out[0]
        ctx.save_for_backward(pts)
        return y, g, s, None

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, y_bar, g_bar, s_bar, tv_bar):
        step = ctx.step
        pts, *tv_saved = ctx.saved_tensors
        n, dev = pts.shape[0], pts.device
        idx, cnt = (None, None) if ctx.sel_index is None else ctx.sel_index
        nl = step.n_levels
        cf, fe, kin = 4 * nl, 20 * nl, 128 + 20 * nl
        fep = step.kp - 128
        npad = (n + 31) // 32 * 32
        f = lambda *shape: torch.empty(*shape, device=dev, dtype=_f32)  # noqa: E731
        lop, rh, re, r0 = f(npad, 4, 6, 128), f(5, npad, 4, 128), f(npad, 4, fep), f(npad, 4, 32)      # point-major operand rows
        f_hat, mu_f, lam_f, w6p = f(npad, cf), f(npad, cf), f(npad, cf), f(npad // 16, step.kp)     # (a row per 16-point workgroup of the launch)
        stash = torch.empty(L.load().gens_sdf_train_stash_bytes(n, 1), device=dev, dtype=torch.uint8)
        cot = [None if t is None else _c(t.to(_f32)) for t in (y_bar, g_bar, s_bar)]
        flops = 8 * 2 * (27 * 128 + (128 + fe) * (4 * 128 + 101 + 1))
        live = None if cnt is None else (cnt, n)
        L.call("gens_sdf_train_bwd", step.packed.table, step.packed.dim_table, nl, step.wf_table, step.wb_table, L.ptr(step.w_last), L.ptr(pts),
               L.ptr(idx, torch.int64), n, L.ptr(cnt, torch.int32), L.ptr(cot[0]), L.ptr(cot[1]), L.ptr(cot[2]), L.ptr(stash, torch.uint8), L.ptr(lop),
               L.ptr(rh), L.ptr(re), L.ptr(r0), L.ptr(f_hat), L.ptr(mu_f), L.ptr(lam_f), L.ptr(w6p), L.stream(),
               nbytes=n * (40 + 4 * (4 * 6 * 128 + 6 * 4 * 128 + 4 * fep + 4 * 32 + 3 * cf)), flops=n * flops, live=live)
        # weight gradients: eleven products over the 4 * npad operand rows in ONE launch (rows of padding points are zero on one side of
        # every product).  Per layer l = 1..5 the hidden columns (lop_l^T rh_l) and the conditioning columns + bias (lop_l^T re) are
        # neighbours in the unit list, so the second product finds lop_l's slab in L2 instead of reading it from HBM again; layer 0 last.
        k = 4 * npad
        fl = 4                                                            # bytes per float
        a_ptr, b_ptr, ldb, ms, ns = [], [], [], [], []
        for l in range(1, 6):
            a_ptr += [lop.data_ptr() + fl * 128 * l] * 2
            b_ptr += [rh.data_ptr() + fl * (l - 1) * k * 128, re.data_ptr()]
            ldb += [128, fep]
            ms += [128, 128]
            ns += [128, fep]
        a_ptr.append(lop.data_ptr())
        b_ptr.append(r0.data_ptr())
        ldb.append(32)
        ms.append(128)
        ns.append(32)
        n_prod = len(ms)
        mi, ni = L.int_table(ms), L.int_table(ns)
        ws = f(L.load().gens_gemm_tn_batch_workspace(n_prod, mi, ni, k))
        sizes = [m * n_ for m, n_ in zip(ms, ns)]
        cc = f(sum(sizes))
        tab = lambda v: C.cast((C.c_void_p * len(v))(*v), C.POINTER(C.c_void_p))  # noqa: E731
        if idx is None:
            L.call("gens_gemm_tn_batch", n_prod, tab(a_ptr), L.int_table([768] * n_prod), tab(b_ptr), L.int_table(ldb), mi, ni, k,
                   L.ptr(ws), L.ptr(cc), L.stream(), nbytes=fl * k * (768 + 32 + 5 * 128 + fep), flops=2 * k * sum(sizes))
        else:       # only the operand rows of the points that exist (32 points -> 128 rows per workgroup of the backward launch)
            L.call("gens_gemm_tn_batch_live", n_prod, tab(a_ptr), L.int_table([768] * n_prod), tab(b_ptr), L.int_table(ldb), mi, ni, k,
                   L.ptr(cnt, torch.int32), 32, 128, L.ptr(ws), L.ptr(cc), L.stream(), nbytes=fl * k * (768 + 32 + 5 * 128 + fep),
                   flops=2 * k * sum(sizes), live=live, label="gens_gemm_tn_batch")
        w6s = w6p.sum(0)
        n_par = ctx.n_par
        if step.raw is not None:
            # d loss / d (weight_v, weight_g, bias) of lin0..lin6 in ONE launch, weight norm's backward included; the 21 gradients are views
            # of one flat buffer (contiguous each: autograd installs them as .grad without a copy)
            sizes_v = [v.numel() for v in step.v]
            rows = [v.shape[0] for v in step.v]
            flat = f(sum(sizes_v) + 2 * sum(rows))
            dv, dg, db, off = [], [], [], 0
            for v in step.v:
                dv.append(flat[off:off + v.numel()].view(v.shape))
                off += v.numel()
            for r in rows:
                dg.append(flat[off:off + r])
                off += r
            for r in rows:
                db.append(flat[off:off + r])
                off += r
            L.call("gens_sdf_train_wgrad", L.ptr_table(step.v), L.ptr_table(step.g), nl, L.ptr(cc), L.ptr(w6s), L.ptr_table(dv), L.ptr_table(dg),
                   L.ptr_table(db), L.stream())
            g_par = [*dv, *[d.view(ctx.shapes[7 + k]) for k, d in enumerate(dg)], *db]
        else:
            parts, off = [], 0
            for m, n_ in zip(ms, ns):
                parts.append(cc[off:off + m * n_].view(m, n_))
                off += m * n_
            w0 = parts[10]
            g_w, g_b = [w0[:, :27]], [w0[:, 27]]
            for l in range(1, 6):
                rows = 101 if l == 2 else 128
                h, e_l = parts[2 * (l - 1)], parts[2 * (l - 1) + 1]
                g_w.append(torch.cat([h, e_l[:, :fe]], 1)[:rows])
                g_b.append(e_l[:rows, fe])
            w6 = torch.zeros(ctx.shapes[6], device=dev, dtype=_f32)
            w6[0] = w6s[:kin]
            b6 = torch.zeros(ctx.shapes[13], device=dev, dtype=_f32)
            b6[0] = w6s[kin]
            g_par = [*g_w, w6, *g_b, b6]
        # volume gradients: the dense TV gradient (when the step carries the regulariser) is WRITTEN first, the look-up's scatter adds into it
        g_vols = [None] * nl
        if any(ctx.needs_input_grad[4 + n_par:]):
            have_tv = bool(tv_saved) and tv_bar is not None
            if have_tv:
                tv_out, *vols = tv_saved
                g_vols = [torch.empty(s, device=dev, dtype=_f32) for s in ctx.shapes[n_par:]]
                L.call("gens_tv_levels_bwd", L.ptr_table(list(vols), align=16), L.ptr_table(step.tv_masks, align=16), L.int_table(ctx.tv_dims), nl,
                       L.ptr(tv_out), L.ptr(_c(tv_bar.detach().to(_f32).reshape(1))), L.ptr_table(g_vols, align=16), L.stream(),
                       nbytes=sum(36 * v[0, 0].numel() for v in vols))
            else:
                g_vols = [torch.zeros(s, device=dev, dtype=_f32) for s in ctx.shapes[n_par:]]
            L.call("gens_sdf_train_scatter", step.packed.dim_table, nl, L.ptr(pts), L.ptr(cot[1]), L.ptr(cot[2]), L.ptr(f_hat), L.ptr(mu_f),
                   L.ptr(lam_f), L.ptr(idx, torch.int64), n, L.ptr(cnt, torch.int32), L.ptr_table(g_vols), L.stream(), nbytes=n * (36 + 3 * 4 * cf),
                   live=live)
        return (None, None, None, None, *g_par, *g_vols)


__all__ = [n_ for n_ in dir() if not n_.startswith("__")]      # private helpers travel too: the package namespace is the old module's
