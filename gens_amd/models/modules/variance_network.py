"""Mirror of models/modules/variance_network.py:5-11: one learnable scalar, inv_s = exp(10 * variance)."""
import torch
import torch.nn as nn


class SingleVarianceNetwork(nn.Module):
    def __init__(self, init_val):
        super().__init__()
        self.variance = nn.Parameter(torch.tensor(init_val))

    def forward(self, x):
        return torch.exp(self.variance * 10.0) * torch.ones(len(x), 1, dtype=x.dtype, device=x.device)
