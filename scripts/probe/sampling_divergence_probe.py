"""Where does a full-size validate() image differ from the CPU oracle once the scene HAS a surface?  Per-ray error quantiles, the same rays
with the hierarchical samples pinned to the device's, and the device's samples against the oracle's own (tests/test_hip_shipped_shapes.py)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gens_amd import ops, synthetic  # noqa: E402
from gens_amd.models.modules.implicit_surface import Scene, reference_jitter  # noqa: E402
from oracle import render_oracle as R  # noqa: E402
from tests.test_hip_shipped_shapes import DIMS5, _scene, _surface  # noqa: E402


def q(t):
    t = t.flatten().float()
    return "median %.2e  p90 %.2e  p99 %.2e  max %.2e  mean %.2e" % tuple(float(x) for x in (t.median(), t.quantile(0.9), t.quantile(0.99), t.max(), t.mean()))


def main(nv, dims, radius, n=128):
    sc = _scene(nv, 480, 640, seed=30, dims=dims)
    surf = _surface(1, dims=dims, radius=radius).cuda().eval()
    ro, rd = synthetic.make_rays(sc["cpu"]["intrs"], sc["cpu"]["c2ws"], 480, 640)
    n_rays = ro.shape[0]
    scene = Scene(sc["vols"], sc["masks"], sc["imgs"], sc["features"], sc["features"], sc["intrs"], sc["c2ws"])
    torch.manual_seed(77)
    jitter = reference_jitter(n_rays)
    g = torch.Generator().manual_seed(3)
    pick = torch.randint(0, n_rays, (n,), generator=g)
    dev = torch.device("cuda")
    ro_c, rd_c = ro[pick].to(dev).contiguous(), rd[pick].to(dev).contiguous()
    with torch.no_grad():
        z0 = ops.coarse_z(sc["near"], sc["far"], surf._coarse_steps(dev), jitter[pick].to(dev), n)
        z_dev = surf._sample_rays(ro_c, rd_c, z0, scene)
        out = surf.render_core(ro_c, rd_c, z_dev, 2.0 / 64, sc["vols"], sc["masks"], sc["features"], sc["features"], sc["imgs"], sc["intrs"], sc["c2ws"],
                               1.0, None, scene=scene, lean=True)
    sd = {k: v.detach().cpu() for k, v in surf.state_dict().items()}
    masks_c = [m.cpu() for m in sc["masks"]]
    cpu = sc["cpu"]
    pr = torch.rand(1024, 3, generator=g) * 2 - 1
    with torch.no_grad():
        z_cpu = R.sample_rays(sd, ro[pick], rd[pick], cpu["near"], cpu["far"], sc["vols_cpu"], masks_c, jitter[pick])
        free = R.render(sd, ro[pick], rd[pick], cpu["near"], cpu["far"], sc["vols_cpu"], masks_c, cpu["imgs"], cpu["features"], cpu["features"],
                        cpu["intrs"], cpu["c2ws"], 1.0, None, jitter[pick], pr, z=z_cpu)
        pinned = R.render(sd, ro[pick], rd[pick], cpu["near"], cpu["far"], sc["vols_cpu"], masks_c, cpu["imgs"], cpu["features"], cpu["features"],
                          cpu["intrs"], cpu["c2ws"], 1.0, None, jitter[pick], pr, z=z_dev.cpu())
    dz = (z_dev.cpu() - z_cpu).abs()
    print("== nv %d, %d levels, radius %.2f, %d rays" % (nv, len(dims), radius, n))
    print("z samples |device - oracle|:", q(dz), "| rays with a sample off by > 1e-4: %d, > 1e-3: %d" % (int((dz.max(1)[0] > 1e-4).sum()), int((dz.max(1)[0] > 1e-3).sum())))
    for name, ref in (("oracle's own samples", free), ("samples pinned to the device's", pinned)):
        col = (out["color_fine"].cpu() - ref["color_fine"]).abs().mean(1)
        dep = (out["render_depth"].cpu().reshape(-1) - ref["render_depth"].reshape(-1)).abs()
        sdd = (out["sdf_depth"].cpu().reshape(-1) - ref["sdf_depth"].reshape(-1)).abs()
        print("  vs %-32s colour %s" % (name, q(col)))
        print("  %-35s depth  %s" % ("", q(dep)))
        print("  %-35s sdf_depth %s" % ("", q(sdd)))
    bad = torch.nonzero(dz.max(1)[0] > 1e-4)[:, 0][:6]
    col = (out["color_fine"].cpu() - free["color_fine"]).abs().mean(1)
    for r in bad.tolist():
        k = int(dz[r].argmax())
        print("  ray %d: worst sample %d, z device %.6f oracle %.6f, colour error %.2e" % (r, k, float(z_dev[r, k]), float(z_cpu[r, k]), float(col[r])))


if __name__ == "__main__":
    main(3, DIMS5, 0.6)
    main(5, [256, 128, 64], 0.97)
