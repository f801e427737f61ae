"""NeRF positional encoding with the reference's layout (models/modules/embedder.py:6-51):
[x, sin(2^0 x), cos(2^0 x), ..., sin(2^(n-1) x), cos(2^(n-1) x)]."""
import torch


class Embedder:
    def __init__(self, input_dims, num_freqs):
        self.freqs = [2.0 ** k for k in range(num_freqs)]
        self.out_dim = input_dims * (1 + 2 * num_freqs)

    def embed(self, x):
        parts = [x]
        for f in self.freqs:
            parts.append(torch.sin(x * f))
            parts.append(torch.cos(x * f))
        return torch.cat(parts, -1)


def get_embedder(multires, input_dims=3):
    e = Embedder(input_dims, multires)
    return e.embed, e.out_dim
