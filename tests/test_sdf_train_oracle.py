"""The two CPU restatements of the training-mode SDF network agree: autograd over the functional MLP with the reference's truncated
sampler (`by_autograd`, pinned to the reference by goldens g17 / g17b / g18 / g18b) and the eight explicit layer sweeps the HIP kernels
gens_sdf_train_{fwd,bwd} execute (`by_sweeps`).  float64, so any disagreement is a wrong formula, not round-off."""
import pytest
import torch

from oracle import sdf_train_oracle as T


def _case(n_levels, n=40, seed=0, dtype=torch.float64):
    g = torch.Generator().manual_seed(seed)
    dims = [9, 6, 5, 4, 3][:n_levels]
    vols = [0.5 * torch.randn(1, 4, d, d, d, generator=g, dtype=dtype) for d in dims]
    W, b = T.shipped_weights(n_levels, seed=seed + 1, scale=1.0, dtype=dtype)
    pts = torch.rand(n, 3, generator=g, dtype=dtype) * 2.4 - 1.2            # some points outside the cube (zero padding)
    pts[0] = 0.0
    cot = [torch.randn(n, k, generator=g, dtype=dtype) for k in (1, 3, 3)]
    return W, b, vols, pts, cot


@pytest.mark.parametrize("n_levels", [1, 2, 3, 4, 5])
def test_sweeps_equal_autograd(n_levels):
    W, b, vols, pts, (yb, gb, sb) = _case(n_levels)
    ref = T.by_autograd(W, b, vols, pts, yb, gb, sb)
    out = T.by_sweeps(W, b, vols, pts, yb, gb, sb)

    def rel(a, c):
        return ((a - c).abs().max() / c.abs().max().clamp_min(1e-30)).item()
    for k in ("y", "g", "s"):
        assert rel(out[k], ref[k]) < 1e-10, k
    for l in range(7):
        assert rel(out["dW"][l][:1] if l == 6 else out["dW"][l], ref["dW"][l][:1] if l == 6 else ref["dW"][l]) < 1e-9, f"dW{l}"
        assert rel(out["db"][l][:1] if l == 6 else out["db"][l], ref["db"][l][:1] if l == 6 else ref["db"][l]) < 1e-9, f"db{l}"
    for i in range(n_levels):
        assert rel(out["dvol"][i], ref["dvol"][i]) < 1e-9, f"dvol{i}"


def test_value_only_cotangent():
    """g_bar = s_bar = 0 (the random / pseudo points of a step, implicit_surface.py:257,490): nu = kappa = rho = 0, omega is the
    ordinary reverse pass."""
    W, b, vols, pts, (yb, gb, sb) = _case(3, n=16, seed=5)
    ref = T.by_autograd(W, b, vols, pts, yb, 0 * gb, 0 * sb)
    out = T.by_sweeps(W, b, vols, pts, yb, 0 * gb, 0 * sb)
    for l in range(6):
        assert torch.allclose(out["dW"][l], ref["dW"][l], rtol=1e-9, atol=1e-12)
    assert torch.allclose(out["dvol"][0], ref["dvol"][0], rtol=1e-9, atol=1e-12)
