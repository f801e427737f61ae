import os, sys, torch
sys.path.insert(0, "/root/repo")
from gens_amd import lib as L, ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
vols = [torch.randn(1, 4, d, d, d, generator=g).to(dev).requires_grad_(True) for d in (256, 128, 64)]
o = (torch.rand(8192, 1, 3, generator=g) * 0.6 - 0.3).to(dev)
d = torch.nn.functional.normalize(torch.randn(8192, 1, 3, generator=g), dim=-1).to(dev)
tt = torch.linspace(-0.9, 0.9, 128, device=dev).view(1, 128, 1)
ray = (o + d * tt).reshape(-1, 3).contiguous()
rand = (torch.rand(1 << 20, 3, generator=g) * 2 - 1).to(dev)
for name, src in (("ray-ordered", ray), ("random", rand)):
    for n in (65536, 98304, 131072, 196608, 262144, 524288):
        row = []
        for least in (10 ** 9, 1):
            ops.kernels.k2_bricks_min = least
            p = src[:n].clone().requires_grad_(True)
            ts = []
            for it in range(10):
                for v in vols:
                    v.grad = None
                f = ops.lookup_volume(p, vols)
                L.profile_begin(only={"gens_lookup_volume_bwd"})
                f.sum().backward()
                ts.append(sum(ms for _, ms, _, _ in L.profile_end(raw=True)) * 1e3)
            ts.sort()
            row.append(ts[len(ts) // 2])
        print(f"{name:12s} n = {n:7d}: direct {row[0]:8.1f} us   bricks {row[1]:8.1f} us")
