"""Host-side model of the image-tile K1 backward's 64-bit fixed-point sums (gens_amd/csrc/k1_volume.hip::tile_add4, volume_build_bwd_tiles_k):
the 1.5 x 2^52 rounding trick, the scale derived from the bits of the call's bound, wrap-around two's-complement sums and the error of a
texel against a float64 sum."""
import numpy as np

MAGIC = 6755399441055744.0          # 1.5 * 2^52
MAGIC_BITS = np.uint64(0x4338000000000000)


def to_fixed(v, w, scale):
    """tile_add4: fma((double)v, (double)w * scale, magic) reinterpreted, minus the bits of magic.  The fma rounds ONCE: exact rational
    arithmetic, then the correctly rounded conversion of Fraction to float."""
    from fractions import Fraction
    ws = np.float64(w) * np.float64(scale)                                       # (one rounding, as on the device)
    d = np.array([float(Fraction(float(a)) * Fraction(float(b)) + Fraction(MAGIC)) for a, b in zip(np.float64(v), ws)], dtype=np.float64)
    return d.view(np.uint64) - MAGIC_BITS


def scales(bound):
    e = (np.float32(bound).view(np.uint32) >> np.uint32(23)) & np.uint32(0xFF)
    scale = np.array((1023 + 40 + 127 - int(e)) << 52, dtype=np.int64).view(np.float64)
    inv = np.array((1023 - 40 - 127 + int(e)) << 52, dtype=np.int64).view(np.float64)
    return float(scale), float(inv)


def test_magic_number_rounds_to_nearest_integer():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(20000) * 2.0 ** rng.integers(-30, 41, 20000), [0.0, 0.5, -0.5, 1.5, -1.5, 2.5, 2.0 ** 41 - 0.25, -(2.0 ** 41), 102819568.5000001]])
    q = to_fixed(x.astype(np.float64), np.ones_like(x), 1.0).view(np.int64)
    assert np.array_equal(q, np.rint(x).astype(np.int64))                        # round half to even, negative values as two's complement


def test_scale_puts_the_bound_between_2_40_and_2_41():
    for bound in (1e-30, 3.3e-7, 0.75, 1.0, 13.8, 1e30, np.finfo(np.float32).max, np.finfo(np.float32).tiny):
        scale, inv = scales(bound)
        assert 2.0 ** 40 <= np.float64(np.float32(bound)) * scale < 2.0 ** 41 and scale * inv == 1.0
    assert ((np.float32(np.inf).view(np.uint32) >> np.uint32(23)) & np.uint32(0xFF)) == 0xFF      # inf / NaN bounds select the float path
    assert ((np.float32(np.nan).view(np.uint32) >> np.uint32(23)) & np.uint32(0xFF)) == 0xFF


def test_a_texel_sum_is_closer_to_the_exact_sum_than_one_float32_rounding():
    rng = np.random.default_rng(1)
    for bound, n in ((13.8, 60000), (2.0e-5, 4096), (7.0e20, 20000)):
        g = (rng.uniform(-1, 1, n) * bound).astype(np.float32)                   # |g| <= bound
        w = rng.uniform(0, 1, n).astype(np.float32) ** 4                         # bilinear weights, many of them tiny
        scale, inv = scales(bound)
        acc = np.uint64(0)
        with np.errstate(over="ignore"):
            acc = np.add.reduce(to_fixed(g, w, scale), dtype=np.uint64)          # wraps like ds_add_u64
        got = np.float32(np.float64(acc.view(np.int64)) * inv)
        exact = float(np.sum(g.astype(np.float64) * w.astype(np.float64)))
        assert abs(float(got) - exact) <= n * 2.0 ** -41 * bound + abs(exact) * 2.0 ** -24
        seq = np.float32(0)
        for a in (g[:2000] * w[:2000]):                                          # what float32 atomics would accumulate (one order of many)
            seq = np.float32(seq + a)
        assert abs(float(got) - exact) <= max(abs(float(seq) - float(np.sum((g[:2000] * w[:2000]).astype(np.float64)))), abs(exact) * 2.0 ** -23) + n * 2.0 ** -41 * bound
