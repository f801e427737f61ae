import os, sys, time, torch, numpy as np
sys.path.insert(0, os.getcwd())
from gens_amd import synthetic, ops
from gens_amd.config import gens_model_conf
from gens_amd.models.modules.implicit_surface import ImplicitSurface, Scene
import gens_amd.models.modules.implicit_surface as M
dev = torch.device("cuda:0")
dims = [256, 128, 64]
sc = synthetic.make_scene(nv=5, h=480, w=640, n_levels=5, seed=0)
imgs, intrs, c2ws = sc["imgs"].to(dev), sc["intrs"].to(dev), sc["c2ws"].to(dev)
feats = [f.to(dev) for f in sc["features"]]
vols = [v.to(dev) for v in synthetic.make_volumes(dims, seed=100)]
ro, rd = synthetic.make_rays(sc["intrs"], sc["c2ws"], 480, 640)
ro, rd = ro.to(dev), rd.to(dev)
near, far = sc["near"].to(dev), sc["far"].to(dev)
torch.manual_seed(0)
surf = ImplicitSurface(gens_model_conf(volume_dims=tuple(dims))["implicit_surface"]).to(dev).eval()
surf.val_chunk = 32768
T = time.perf_counter
import types
orig_render = surf.render
marks = []
def render(*a, **k):
    r = orig_render(*a, **k)
    marks.append(T())
    return r
surf.render = render
for it in range(3):
    marks.clear()
    torch.cuda.synchronize(); t0 = T()
    with torch.no_grad():
        _, masks = ops.volume_build(feats[:3], intrs, c2ws, dims)
        scene = Scene(vols, masks, imgs, feats, feats, intrs, c2ws)
        t1 = T()
        out = surf.validate(ro, rd, near, far, vols, masks, imgs, feats, feats, intrs, c2ws, None, None, (480, 640), extract_geometry=False, scene=scene)
        t2 = T()
    torch.cuda.synchronize(); t3 = T()
    print(f"setup(host) {1e3*(t1-t0):.1f}  validate {1e3*(t2-t1):.1f}  first render returned +{1e3*(marks[0]-t1):.1f}  last render returned +{1e3*(marks[-1]-t1):.1f}  tail after last render {1e3*(t2-marks[-1]):.1f}  total {1e3*(t3-t0):.1f}")
