"""NeRF positional encoding with the reference's layout (models/modules/embedder.py:6-51):
[x, sin(2^0 x), cos(2^0 x), ..., sin(2^(n-1) x), cos(2^(n-1) x)].

Same values as the reference's per-frequency lambdas (x * 2^k is exact, sin / cos are the same element-wise kernels), but all
frequencies go through ONE multiply, ONE sin and ONE cos: 5 launches instead of 4 n + 1, and as many fewer autograd nodes in the first,
second and third derivative the training step takes through it (it is launch-bound: DESIGN.md section 8)."""
import torch


class Embedder:
    def __init__(self, input_dims, num_freqs):
        self.freqs = [2.0 ** k for k in range(num_freqs)]
        self.out_dim = input_dims * (1 + 2 * num_freqs)
        self._freq_cache = {}

    def _freq_tensor(self, x):
        key = (x.device, x.dtype)
        f = self._freq_cache.get(key)
        if f is None:                      # exact powers of two, uploaded once per device (not once per call)
            f = torch.tensor(self.freqs, device=x.device, dtype=x.dtype)[:, None]
            self._freq_cache[key] = f
        return f

    def embed(self, x):
        if not self.freqs:
            return x
        xf = x.unsqueeze(-2) * self._freq_tensor(x)                       # (..., n, d)
        sc = torch.stack([torch.sin(xf), torch.cos(xf)], dim=-2)          # (..., n, 2, d): sin block, then cos block, per frequency
        return torch.cat([x, sc.reshape(*x.shape[:-1], -1)], -1)


def get_embedder(multires, input_dims=3):
    e = Embedder(input_dims, multires)
    return e.embed, e.out_dim
