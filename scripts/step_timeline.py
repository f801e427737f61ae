import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gens_amd import ops, synthetic
from gens_amd.config import gens_model_conf
from gens_amd.models.modules.implicit_surface import ImplicitSurface, Scene, JitterStream
from gens_amd.models.modules.volume import Volume
dev = torch.device("cuda:0")
dims=[256,128,64]
sc = synthetic.make_scene(nv=5, h=480, w=640, n_levels=5, seed=0)
imgs, intrs, c2ws = sc["imgs"].to(dev), sc["intrs"].to(dev), sc["c2ws"].to(dev)
feats = [f.to(dev) for f in sc["features"]]; near, far = sc["near"].to(dev), sc["far"].to(dev)
vols = [v.to(dev) for v in synthetic.make_volumes(dims, seed=100)]
ro, rd = synthetic.make_rays(sc["intrs"], sc["c2ws"], 480, 640); ro, rd = ro.to(dev), rd.to(dev)
torch.manual_seed(0)
conf = gens_model_conf(volume_dims=tuple(dims))
surf = ImplicitSurface(conf["implicit_surface"]).to(dev).eval(); surf.val_chunk = 32768
volume = Volume(conf["volume"])
def T():
    torch.cuda.synchronize(); return time.perf_counter()
for it in range(3):
    with torch.no_grad():
        t0=T(); cost, masks = volume.agg_mean_var(feats, intrs, c2ws); t1=T()
        scene = Scene(vols, masks, imgs, feats, feats, intrs, c2ws); scene.volumes_nograd(); t2=T()
        js = JitterStream(ro.shape[0], 32768); first = js.slice(0, 32768); t3=T()
        n = ro.shape[0]; outs=[]
        for s in range(0, n, 32768):
            e=min(s+32768,n)
            outs.append(surf.render(ro[s:e], rd[s:e], near, far, vols, masks, imgs, feats, feats, intrs, c2ws, 1.0, None, scene=scene, lean=True, t_rand=js.slice(s,e)))
        t4h=time.perf_counter(); t4=T()
        col = torch.cat([o["color_fine"] for o in outs]).cpu(); t5=T()
    if it: print(f"K1 {1e3*(t1-t0):.1f}  scene/pack {1e3*(t2-t1):.1f}  jitter-first {1e3*(t3-t2):.1f}  render loop {1e3*(t4-t3):.1f} (host enqueue done after {1e3*(t4h-t3):.1f})  d2h {1e3*(t5-t4):.1f}")
for it in range(2):
    with torch.no_grad():
        cost, masks = volume.agg_mean_var(feats, intrs, c2ws)
        scene = Scene(vols, masks, imgs, feats, feats, intrs, c2ws)
        t0 = T()
        out = surf.validate(ro, rd, near, far, vols, masks, imgs, feats, feats, intrs, c2ws, None, None, (480, 640), extract_geometry=False, scene=scene)
        t1 = T()
    print(f"validate {1e3*(t1-t0):.1f} ms")
