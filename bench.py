#!/usr/bin/env python3
"""Benchmark of the GenS hot path on MI355X: SDF ray-samples / second (BASELINE.json metric).

One step = one scene of BASELINE config[1] ("5-view 480x640, full 3-scale volumes, inference"):
    K1  cost-volume build for the 5-view feature pyramid, volume_dims = [256, 128, 64]
    +   rendering of every pixel ray of the reference view (307 200 rays x 128 final samples) through
        ImplicitSurface.validate's path: hierarchical sampling (4 rounds), SDF MLP + first derivatives,
        source-view feature look-up, blending MLP, compositing -> colour / depth / normal / SDF-depth buffers.
Inputs are synthetic (gens_amd/synthetic.py: DTU-like cameras, random images / features, seeded geometric-init
MLPs) and are resident in HBM before the timed region.  The 2-D CNN and the 3-D U-Net are outside the accelerated
path: the renderer reads a synthetic stand-in for the regularised volumes, while K1 still
runs inside every timed step and supplies the visibility masks.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python bench.py --gpus N                                                   (starts its own N rank processes: launch_ranks below)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N   (one scene per rank, weak scaling; --gpus must equal WORLD_SIZE)

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant HIP kernel, HIP-event
timed inside the timed region) and `cpu_baseline` (the CPU oracle on a bounded ray sample, rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

# torch is imported by main() of a RANK process only: the launching parent of `--gpus N` (launch_ranks) starts its children and relays
# rank 0's line without ever loading torch, let alone touching HIP
torch = None

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
F32_MFMA_PEAK_TFLOPS = 157.3   # dense fp32-input MFMA peak (same guide)
F16_MFMA_PEAK_TFLOPS = 2500.0  # dense f16/bf16 MFMA peak (same guide)
DOMINANT = "gens_sdf_mlp:grad"  # the kernel the roofline object is about: the fwd + d/dx launches of the fused SDF network, device kernel
                                # sdf_mlp_k<FE, true> (asserted against the measured table).  The value-only passes run the transposed
                                # kernel sdf_value_t_k (profile key "gens_sdf_value"), listed beside it; the secondary workloads never launch the
                                # GRAD kernel at other shapes, so rocprof's per-process average of that symbol is the timed region's


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--rays", type=int, default=480 * 640, help="rays per step and rank (default: the full 480x640 image)")
    p.add_argument("--chunk", type=int, default=None, help="rays per render() chunk (default: what ImplicitSurface.validate chooses itself -- equal "
                                                           "chunks of at most 32 768 rays; the bench sets nothing on the model)")
    p.add_argument("--dims", type=int, nargs="+", default=[256, 128, 64])
    p.add_argument("--views", type=int, default=5)
    p.add_argument("--cpu-rays", type=int, default=640, help="rays of the CPU-oracle baseline sample (0 = skip); ~15 s on a 128-core host")
    p.add_argument("--no-kernel-timing", action="store_true")
    p.add_argument("--shard", default="scenes", choices=["scenes", "rays"],
                   help="N > 1: 'scenes' = one scene per rank (weak scaling, the headline the driver runs); 'rays' = ONE scene whose rays are "
                        "split across the ranks (BASELINE config 4: strong scaling), rendered buffers gathered over RCCL inside the timed region")
    p.add_argument("--headline-only", action="store_true", help="skip the secondary figures (training steps, validation item, five-level and "
                                                                "split-half variants) that follow the timed region at N = 1")
    p.add_argument("--train-step", action="store_true", help=argparse.SUPPRESS)      # round-1 flag: the training figures are on by default now
    p.add_argument("--sdf-precision", default="f32", choices=["f32", "f16x2"],
                   help="f32: exact float32 MFMA (headline); f16x2: split-half operands on the f16 matrix cores (~1e-6 relative)")
    return p.parse_args()


def gpu_sysfs_dir(device_index=0):
    """/sys/class/drm/cardN/device of the GPU this process computes on.  A box exposes EVERY GPU of the node in sysfs, whatever the process
    may use: the card is matched by its PCI address (torch's device properties; no HIP call beyond what the bench already made)."""
    import glob
    try:
        pr = torch.cuda.get_device_properties(device_index)
        want = "%04x:%02x:%02x" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
    except Exception:
        return None
    for path in sorted(glob.glob("/sys/class/drm/card*/device")):
        if os.path.basename(os.path.realpath(path)).lower().startswith(want):
            return path
    return None


def gpu_clocks(card_dir):
    """One reading of the card's hwmon: shader / memory clock (MHz), socket power (W), busy percent; {} where the files are not readable."""
    import glob
    out = {}
    if not card_dir:
        return out
    for hw in glob.glob(os.path.join(card_dir, "hwmon", "hwmon*")):
        for key, name, scale in (("sclk_mhz", "freq1_input", 1e-6), ("mclk_mhz", "freq2_input", 1e-6), ("power_w", "power1_input", 1e-6)):
            try:
                out[key] = int(round(int(open(os.path.join(hw, name)).read()) * scale))
            except (OSError, ValueError):
                pass
    try:
        out["busy_percent"] = int(open(os.path.join(card_dir, "gpu_busy_percent")).read())
    except (OSError, ValueError):
        pass
    return out


from gens_amd.chunking import balanced_chunk  # noqa: E402  (torch-free; what ImplicitSurface.validate uses when val_chunk is left alone)


class ClockSampler:
    """Shader / memory clock, power and busy readings from sysfs WHILE the timed region runs (a helper thread, one read every `period` seconds; no child
    process): a reading taken before or after the region sees an idle GPU and says nothing about it."""

    def __init__(self, device_index=0, period=0.05):
        import threading
        self.card = gpu_sysfs_dir(device_index)
        self.period, self.samples, self._stop = period, [], threading.Event()
        self.thread = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        while not self._stop.is_set():
            c = gpu_clocks(self.card)
            if c:
                self.samples.append(c)
            self._stop.wait(self.period)

    def __enter__(self):
        self.thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        self.thread.join()

    def summary(self):
        out = {"samples": len(self.samples), "where": "%s/hwmon (this process' GPU by PCI address), sampled inside the timed region" % self.card}
        for key in ("sclk_mhz", "mclk_mhz", "power_w", "busy_percent"):
            v = sorted(c[key] for c in self.samples if key in c)
            if v:
                out[key] = {"min": v[0], "median": v[len(v) // 2], "max": v[-1]}
        return out


SECONDARY_STEPS = 5      # timed steps of every secondary figure (the headline takes --steps)


def run_timed(step, n=SECONDARY_STEPS):
    """n steps, each timed on the host (every step here ends with validate()'s stream synchronisation) -> (mean seconds, per-step ms)."""
    ms = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        t1 = time.perf_counter()
        step()
        ms.append((time.perf_counter() - t1) * 1e3)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, ms


def percentiles(ms):
    """median / p10 / p90 of per-step wall times (SURVEY section 8d's protocol)."""
    s = sorted(ms)
    pick = lambda q: s[min(len(s) - 1, max(0, int(round(q * (len(s) - 1)))))]  # noqa: E731
    return {"median": round(pick(0.5), 3), "p10": round(pick(0.1), 3), "p90": round(pick(0.9), 3), "min": round(s[0], 3), "max": round(s[-1], 3)}


def build_model(dims, device):
    from gens_amd.config import gens_model_conf
    from gens_amd.models.modules.implicit_surface import ImplicitSurface
    from gens_amd.models.modules.volume import Volume
    torch.manual_seed(0)
    conf = gens_model_conf(volume_dims=tuple(dims), n_feature_levels=5)
    surf = ImplicitSurface(conf["implicit_surface"]).to(device).eval()
    vol = Volume(conf["volume"])
    return surf, vol


def launch_ranks(n, argv, env=None, child=None, poll_s=0.05):
    """`bench.py --gpus N` without a launcher around it: start N fresh rank processes of this file (what the reference's scripts/run.sh:3 does
    with torch.distributed.launch and utils/distribute.py:66-88 reads back from the environment), one per GPU, rendezvous on 127.0.0.1.

    The parent never initialises HIP (it does not even import torch): it starts the children, relays rank 0's stdout -- the ONE JSON line --
    to its own stdout, sends the other ranks' stdout to stderr, and returns the first non-zero exit code.  A rank that dies takes the others
    down with it (they would otherwise wait in a collective for ever): the exact child processes are terminated, nothing is retried.

    child: the command of a rank process (tests substitute a stub); default: this interpreter on this file with the same arguments."""
    import shutil
    import signal
    import socket
    import subprocess
    import tempfile
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:        # a free port for MASTER_PORT (what the environment contract names) ...
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    # ... but the rendezvous itself goes through a FILE store in a directory of our own: between closing the socket above and rank 0's bind
    # another process could take the port, a file nobody else knows cannot be taken
    store_dir = tempfile.mkdtemp(prefix="gens_bench_rdzv_")
    cmd = list(child) if child is not None else [sys.executable, os.path.abspath(__file__)] + list(argv)
    base = dict(os.environ if env is None else env)
    base.update({"WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                 "GENS_BENCH_INIT": "file://" + os.path.join(store_dir, "store")})
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")                      # dmabuf IPC: what RCCL needs on this driver
    base.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    procs = []

    def stop_children(grace_s=5.0):
        """terminate(), then kill() what is still alive: these exact children, never a pattern."""
        alive = [p_ for p_ in procs if p_.poll() is None]
        for p_ in alive:
            p_.terminate()
        deadline = time.time() + grace_s
        for p_ in alive:
            try:
                p_.wait(max(0.0, deadline - time.time()))
            except subprocess.TimeoutExpired:
                p_.kill()
                p_.wait()

    class _Stopped(Exception):
        pass

    def on_term(signum, frame):                                             # a harness timeout sends SIGTERM: leave through the finally below
        raise _Stopped(signum)

    previous = None
    try:
        previous = signal.signal(signal.SIGTERM, on_term)
    except ValueError:                                                      # (not the main thread: tests call this from wherever they like)
        previous = None
    import threading
    chunks = []
    rc = 0
    try:
        for r in range(n):
            e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
            procs.append(subprocess.Popen(cmd, env=e, stdout=subprocess.PIPE if r == 0 else sys.stderr.fileno()))
        # rank 0 writes its line once, at the end: reading its pipe on a helper thread keeps a chatty library from filling it
        reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
        reader.start()
        live = set(range(n))
        while live:
            for r in sorted(live):
                code = procs[r].poll()
                if code is None:
                    continue
                live.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    sys.stderr.write("bench.py: rank %d exited with status %d; stopping the other ranks\n" % (r, code))
                    stop_children()                                         # (they would wait in a collective for ever)
            if live:
                time.sleep(poll_s)
    except (KeyboardInterrupt, _Stopped) as stop:                           # the PARENT was interrupted: nothing may stay behind holding a GPU
        sys.stderr.write("bench.py: launcher interrupted (%r); stopping %d rank processes\n" % (stop, sum(p_.poll() is None for p_ in procs)))
        rc = 130 if isinstance(stop, KeyboardInterrupt) else 143
    finally:
        stop_children()
        if previous is not None:
            signal.signal(signal.SIGTERM, previous)
        shutil.rmtree(store_dir, ignore_errors=True)
    if procs and procs[0].stdout is not None:
        reader.join(10)
    out = b"".join(chunks)
    if rc == 0:
        os.write(1, out)
    elif out:
        os.write(2, out)
    return rc


def main():
    global DOMINANT, torch
    args = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        # no launcher around us: become the launcher.  Nothing above this line has loaded torch or HIP, and nothing below it runs in the parent.
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    if env_world is not None and int(env_world) != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%s: the launcher's --nproc-per-node and --gpus must agree" % (args.gpus, env_world))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")                    # dmabuf IPC: what RCCL needs on this driver (also under torch.distributed.run)
    import torch
    # The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints its version banner through the C stdio buffer, which is flushed
    # at exit, i.e. after the line): everything else that goes to file descriptor 1 is sent to stderr, the line is written to the real stdout.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    if len(args.dims) in (3, 5) and os.environ.get("GENS_SDF_GRAD_ROWMAJOR") is None:
        DOMINANT = "gens_sdf_grad"          # the transposed value + gradient kernel sdf_grad_t_k (k6g_sdf_grad.hip), float32
    else:                                   # under either --sdf-precision (the split-half arithmetic covers the value-only passes only)
        DOMINANT = "gens_sdf_mlp:grad"
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    if os.environ.get("GENS_BENCH_ONE_DEVICE"):       # test aid: every rank on GPU 0 (a 2-rank dry run on a 1-GPU box, with GENS_BENCH_BACKEND=gloo)
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1 or os.environ.get("GENS_BENCH_FORCE_DIST"):      # the env switch exercises the RCCL path on a 1-GPU box
        import torch.distributed as dist
        # 'nccl' is RCCL on ROCm.  Under our own launcher the rendezvous is a file store (GENS_BENCH_INIT, see launch_ranks); under
        # torch.distributed.run it is the launcher's env:// contract
        init = os.environ.get("GENS_BENCH_INIT", "env://")
        dist.init_process_group(backend=os.environ.get("GENS_BENCH_BACKEND", "nccl"), init_method=init, world_size=world, rank=rank)

    from gens_amd import lib as L
    from gens_amd import synthetic
    from gens_amd.models.modules.implicit_surface import Scene
    L.load()

    h, w = 480, 640
    by_rays = args.shard == "rays" and dist is not None
    seed = 0 if by_rays else rank                                                  # 'rays': every rank holds the SAME scene
    sc = synthetic.make_scene(nv=args.views, h=h, w=w, n_levels=5, seed=seed)      # 'scenes': one scene per rank
    imgs, intrs, c2ws = sc["imgs"].to(dev), sc["intrs"].to(dev), sc["c2ws"].to(dev)
    feats = [f.to(dev) for f in sc["features"]]
    near, far = sc["near"].to(dev), sc["far"].to(dev)
    vols = [v.to(dev) for v in synthetic.make_volumes(args.dims, seed=100 + seed)]
    rays_o, rays_d = synthetic.make_rays(sc["intrs"], sc["c2ws"], h, w)
    rays_o, rays_d = rays_o[:args.rays].to(dev), rays_d[:args.rays].to(dev)
    n_rays = rays_o.shape[0]
    surf, volume = build_model(args.dims, dev)
    if args.chunk:                      # (the default run leaves the model as its constructor built it: the measured configuration IS the shipped one)
        surf.val_chunk = args.chunk
    if args.sdf_precision != "f32":
        surf.sdf_precision = args.sdf_precision
    n_final = surf.n_samples + surf.n_importance
    hw = (1, n_rays)

    state = {}
    shard = None
    if by_rays:
        from gens_amd.distributed import Shard
        shard = Shard()
        r0, r1 = shard.rays(n_rays)       # (validate() cuts THIS rank's ray range into equal chunks itself)

    def step():
        # the previous step's 690 MB of cost volumes and masks go back to the allocator BEFORE this step's are built (no second set of segments)
        state.pop("cost", None)
        state.pop("masks", None)
        with torch.no_grad():
            cost_volumes, masks = volume.agg_mean_var(feats, intrs, c2ws)                # K1 (cost volumes feed the U-Net upstream)
            scene = Scene(vols, masks, imgs, feats, feats, intrs, c2ws)
            out = surf.validate(rays_o, rays_d, near, far, vols, masks, imgs, feats, feats, intrs, c2ws, None, None, hw,
                                extract_geometry=False, scene=scene, shard=shard)       # shard: this rank's ray range + RCCL gather inside
            # (image after image, as a validation loop renders them: validate() itself starts the NEXT image's jitter draws on a helper thread
            # before it waits for this image's copy -- ImplicitSurface._speculate_jitter; rounds 2 - 5 called surf.prefetch_jitter() here)
        if dist is not None and not by_rays:                                             # config 4: gather of rendered buffers
            buf = surf.last_device_image                                                 # (P, 8) rgb | normal | sdf depth | rendered depth, on the device
            gathered = torch.empty(world * buf.shape[0], buf.shape[1], device=dev, dtype=buf.dtype)
            if dist.get_backend() == "nccl":
                dist.all_gather_into_tensor(gathered, buf)                               # RCCL over xGMI, device to device
            else:                                                                        # (gloo dry run)
                dist.all_gather(list(gathered.view(world, *buf.shape).unbind(0)), buf)
            state["gathered"] = gathered
        state["out"], state["masks"], state["cost"] = out, masks, cost_volumes

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    torch.manual_seed(1234)
    for _ in range(args.warmup):
        step()
    sync()
    # The first step of a process creates ~10^5 long-lived Python objects (modules, plans, cached tensors); the cyclic collector's next full
    # pass over them (~35 ms) otherwise lands inside the second image, while the queue to the GPU is still short.  Collect now and move the
    # survivors out of the collector's way (what `timeit` does more bluntly by disabling the collector).
    import gc
    gc.collect()
    gc.freeze()
    # HIP events bracket the launches of the dominant kernel inside the timed region (roofline.achieved); the table of all
    # kernels comes from one extra, untimed step so that ~600 event records per step do not sit in the measured time
    K1 = "gens_volume_build_levels"         # the north star's own kernel: one launch per step, HIP-event timed inside the timed region like DOMINANT
    if not args.no_kernel_timing:
        L.profile_begin(only={DOMINANT, K1})
    step_ms = []
    clocks = ClockSampler(local)
    with clocks:
        t0 = time.perf_counter()
        for _ in range(args.steps):
            t_step = time.perf_counter()
            step()                           # (validate() ends with the image's device-to-host copy: a step's wall time is its own)
            step_ms.append((time.perf_counter() - t_step) * 1e3)
        sync()
        elapsed = time.perf_counter() - t0
    kernels, each_launch = L.profile_end(per_launch=True) if not args.no_kernel_timing else ({}, {})
    headline_chunk = getattr(surf, "last_val_chunk", args.chunk)          # (the secondaries below render other ray counts with the same model)
    timed_steps = {k: args.steps for k in kernels}
    if not args.no_kernel_timing:
        L.profile_begin()
        step()
        sync()
        for name, k in L.profile_end().items():
            if name not in kernels:
                kernels[name], timed_steps[name] = k, 1
    rank_ms = [elapsed / args.steps * 1e3]
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)                                                        # each rank's own time: the line reports min / max
        rank_ms = [float(x) / args.steps * 1e3 for x in every]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)

    # secondary figure (N > 1, headline = one scene per rank): the SAME run's strong-scaling number -- ONE scene whose rays are split
    # across the ranks (BASELINE config 4), gather inside the timed region.  Every rank takes part (collectives), rank 0 reports.
    ray_sharded = None
    if dist is not None and (world > 1 or os.environ.get("GENS_BENCH_FORCE_DIST") == "2") and not by_rays and not args.headline_only:
        try:
            ray_sharded = ray_sharded_variant(args, dev, dist, surf, volume, n_final, sync)
        except Exception as e:
            ray_sharded = {"error": f"{type(e).__name__}: {e}"}

    def finish():
        surf.join_speculation()          # the draws for an image that will not be rendered: let the helper thread finish before the interpreter tears down
        if dist is not None:
            dist.destroy_process_group()

    if rank != 0:
        finish()
        return

    total_ray_samples = (1 if by_rays else world) * n_rays * n_final * args.steps
    value = total_ray_samples / elapsed
    # sanity of the rendered buffers (a bench that renders garbage is not a bench)
    col = state["out"]["color_fine"]
    assert torch.isfinite(col).all() and float(state["masks"][0].mean()) > 0.01

    # K1 as the STEP pays for it (after a render: cold L2 / MALL, outputs taken from the allocator): every launch of the timed region
    k1_in_step = None
    if each_launch.get(K1):
        in_order = [round(x * 1e3, 1) for x in each_launch[K1]]
        ms = sorted(each_launch[K1])
        k1_bytes = kernels[K1]["bytes"] / kernels[K1]["launches"]
        med = ms[len(ms) // 2]
        k1_in_step = {"kernel": K1, "launches": len(ms), "median_us": round(med * 1e3, 1), "p10_us": round(ms[len(ms) // 10] * 1e3, 1),
                      "p90_us": round(ms[(9 * len(ms)) // 10] * 1e3, 1), "algorithmic_bytes_per_launch": int(k1_bytes),
                      "achieved_GBs": round(k1_bytes / 1e9 / (med / 1e3), 1), "peak_GBs": HBM_PEAK_GBS, "hbm_frac": round(k1_bytes / 1e9 / (med / 1e3) / HBM_PEAK_GBS, 4),
                      "us_in_launch_order": in_order,
                      "measured": "HIP events around every launch inside the timed region (one per step, all levels of the scene); north star: >= 0.40"}
    roofline = None
    table = {}
    if kernels:
        hip_ms = sum(k["ms"] / timed_steps[n] for n, k in kernels.items()) * args.steps
        for name, k in sorted(kernels.items(), key=lambda kv: -kv[1]["ms"] / timed_steps[kv[0]]):
            table[name] = {"launches": k["launches"], "ms_per_step": round(k["ms"] / timed_steps[name], 3),
                           "measured": "timed region" if name in (DOMINANT, K1) else "one extra untimed step",
                           "algo_GBs": round(k["bytes"] / 1e9 / (k["ms"] / 1e3), 1) if k["ms"] > 0 and k["bytes"] else None}
        for name, k in kernels.items():
            if k.get("flops"):
                table[name]["TFLOPs"] = round(k["flops"] / 1e12 / (k["ms"] / 1e3), 1)
        # the roofline object is about the kernel whose launches were HIP-event timed INSIDE the timed region (DOMINANT); whether it is also
        # the slowest one of the untimed extra step is reported, not asserted (a surprise must not cost an N-rank run its line)
        slowest = max(((n, k) for n, k in kernels.items() if k["bytes"]), key=lambda kv: kv[1]["ms"] / timed_steps[kv[0]])[0]
        dom_name, dom = (DOMINANT, kernels[DOMINANT]) if DOMINANT in kernels else (slowest, kernels[slowest])
        # HBM bytes per launch of the dominant entry point: from the newest committed PMC passes of THIS command (two separate rocprofv3
        # --pmc runs, scripts/pmc_traffic.py); the file it came from is named beside the number
        traffic = traffic_source = None
        import glob
        # ... and only if that file was taken with the kernel sources of THIS tree (their hash is stored beside the counters): a stale
        # number is dropped, not reported
        from scripts.pmc_traffic import kernel_source_hash
        for tpath in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
            doc = json.load(open(tpath))
            t = doc["kernels"].get(dom_name.split(":")[0])
            if t and args.chunk in (None, 32768) and args.views == 5 and doc.get("kernel_source_hash") == kernel_source_hash():
                traffic = t["traffic_bytes_per_launch_corrected"]      # 2 x FETCH_SIZE + WRITE_SIZE (gfx950 correction of the guide)
                traffic_source = "profiles/" + os.path.basename(tpath) + " (mean over the entry point's launches)"
                break
        common = {"kernel": dom_name, "slowest_kernel_of_the_step": slowest, "traffic": traffic, "traffic_source": traffic_source,
                  "traffic_note": ("gens_sdf_grad parks softplus' of one layer (16 KB per 32 points) in a SIMD-private slot of global memory between the "
                                   "forward and the reverse chain: the slots (2 MB per XCD) are read back from L2, but every store still reaches the fabric "
                                   "(WRITE_SIZE 2.2 GB per launch; as compiler-private memory the reads missed too: 4.8 GB) on top of ~0.15 GB of "
                                   "algorithmic bytes; a matrix-pipe-bound kernel (DESIGN.md section 4b')") if dom_name == "gens_sdf_grad" else None,
                  "avg_launch_us": round(dom["ms"] * 1e3 / dom["launches"], 2),
                  "hip_kernels_ms_per_step": round(hip_ms / args.steps, 2)}
        if dom.get("flops"):      # the fused MLP is matrix-core bound: price it against the dense MFMA peak of the operand type
            split = args.sdf_precision != "f32"     # split-half: every fp32 product costs three f16 products on the f16 pipe
            peak = F16_MFMA_PEAK_TFLOPS if split else F32_MFMA_PEAK_TFLOPS
            achieved = (3 if split else 1) * dom["flops"] / 1e12 / (dom["ms"] / 1e3)
            roofline = {"bound": "mfma", "achieved": round(achieved, 1), "peak": peak, "unit": "TFLOP/s",
                        "frac": round(achieved / peak, 4), "algorithmic_flops_per_launch": int((3 if split else 1) * dom["flops"] / dom["launches"]),
                        **common}
        else:
            achieved = dom["bytes"] / 1e9 / (dom["ms"] / 1e3)
            roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": int(dom["bytes"] / dom["launches"]), **common}
        # the dominant HBM-bound gather kernel is reported alongside (north-star target: >= 40 % of HBM peak)
        hbm_name, hbm = max(((n, k) for n, k in kernels.items() if k["bytes"] and not k.get("flops")), key=lambda kv: kv[1]["ms"] / timed_steps[kv[0]])
        hb = hbm["bytes"] / 1e9 / (hbm["ms"] / 1e3)
        roofline["dominant_hbm_kernel"] = {"kernel": hbm_name, "achieved": round(hb, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                           "frac": round(hb / HBM_PEAK_GBS, 4), "avg_launch_us": round(hbm["ms"] * 1e3 / hbm["launches"], 2)}

    # secondary figure (N = 1, default precision only): the same step with the opt-in split-half SDF kernel (every fp32 operand as an
    # f16 hi + lo pair, three f16 MFMAs per product, fp32 accumulate; ~1e-6 relative to the fp32 kernel).
    # Reported beside the headline, never as `value`.
    split = None
    if world == 1 and args.sdf_precision == "f32" and not args.no_kernel_timing and not args.headline_only:
        surf.sdf_precision = "f16x2"
        step()
        sync()
        dt, split_ms = run_timed(step)
        L.profile_begin()                          # the C-ABI kernel table of one extra step in this arithmetic (HIP events per launch)
        step()
        sync()
        split_rows = kernel_rows(L.profile_end(), 4)
        for name_, row_ in split_rows.items():     # f16 products issued: three per float32 product, against the dense f16 peak
            if name_.endswith("_f16") and "TFLOPs" in row_:
                row_["f16_pipe_frac"] = round(3 * row_["TFLOPs"] / F16_MFMA_PEAK_TFLOPS, 4)
        # what the arithmetic costs in the image: the same rays without jitter under both settings, L1 of colour and rendered depth
        perturb, surf.perturb = surf.perturb, 0
        images = {}
        for prec in ("f32", "f16x2"):
            surf.sdf_precision = prec
            step()
            sync()
            images[prec] = surf.last_device_image.clone()
        surf.perturb, surf.sdf_precision = perturb, "f32"
        diff = (images["f16x2"] - images["f32"]).abs()
        split = {"sdf_precision": "f16x2", "value": n_rays * n_final / dt, "unit": "ray-samples/s", "ms_per_step": dt * 1e3, "steps": SECONDARY_STEPS,
                 "ms_per_step_stats": percentiles(split_ms), "colour_L1_vs_f32": float(diff[:, 0:3].mean()), "depth_L1_vs_f32": float(diff[:, 7].mean()),
                 "depth_max_abs_vs_f32": float(diff[:, 7].max()),
                 "rays_with_depth_moved_by_more_than_1e-4": int((diff[:, 7] > 1e-4).sum()), "rays": int(diff.shape[0]), "hip_kernels": split_rows,
                 "note": "opt-in: every pass of the SDF network in split-half f16 arithmetic -- the value-only passes of the hierarchical "
                         "sampling on gens_sdf_value_f16, render_core's value + gradient pass on gens_sdf_grad_f16 (operands as f16 hi + lo "
                         "pairs, three f16 MFMAs per product, float32 accumulation, softplus' in float32); not the headline"}

    cpu = None
    if world == 1 and args.cpu_rays > 0:
        cpu = cpu_baseline(args, surf, sc, vols, state["masks"], n_final)

    secondary = world == 1 and not args.headline_only
    # secondary figure (N = 1): what one rank of an 8-way ray-sharded image (BASELINE config[3]) has to do, timed alone on this GPU
    projection = None
    if secondary and dist is None and args.sdf_precision == "f32":
        try:
            projection = ray_sharded_variant(args, dev, None, surf, volume, n_final, torch.cuda.synchronize, projection=(7, 8))
            projection["implied_speedup_upper_bound"] = round(elapsed / args.steps * 1e3 / projection["ms"], 2)
            projection["note"] = ("implied_speedup_upper_bound = the headline's ms_per_step / ms: the most an 8-GPU ray-sharded image can gain if the gather "
                                  "and the slowest rank cost nothing extra; the K1 build and the per-image host work do not shrink with the ray count")
        except Exception as e:
            projection = {"error": f"{type(e).__name__}: {e}"}
    # secondary figure (N = 1): the headline step at the SHIPPED level count (confs/gens.conf:63-67: five volume levels, sdf_mlp_k<100>)
    levels5 = views3 = None
    if secondary and args.sdf_precision == "f32" and len(args.dims) == 3:
        try:
            levels5 = five_level_variant(args, dev, sc, feats, imgs, intrs, c2ws, near, far, rays_o, rays_d, n_final)
        except Exception as e:
            levels5 = {"error": f"{type(e).__name__}: {e}"}
        # ... and the DTU TEST protocol as shipped (confs/gens.conf:17-30, BASELINE config[3] on one GPU): num_src_view = 2 -> three views,
        # five levels; the blending kernel runs its S = 2 instantiation
        try:
            views3 = five_level_variant(args, dev, sc, [f[:3].contiguous() for f in feats], imgs[:3].contiguous(), intrs[:3].contiguous(),
                                        c2ws[:3].contiguous(), near, far, rays_o, rays_d, n_final, kernels=True)
            views3["views"] = 3
            views3["note"] = ("the shipped test protocol: num_src_view = 2 (three views), five volume levels; the blending kernel's S = 2 "
                              "instantiation (gens_blend_views_t); not the headline")
        except Exception as e:
            views3 = {"error": f"{type(e).__name__}: {e}"}

    # secondary figures (N = 1): TRAINING steps of BASELINE config[2] / config[4] shape -- 5 views 480x640, volume_dims [256, 128, 64], 512 rays
    # + 2048 pseudo points, a reference-like loss, backward through every kernel, Adam (scripts/train_step_bench.py):
    #   "full"      GenS.forward as runner.py runs it: 2-D feature CNN (twice), volume build, 3-D U-Net, render
    #   "hot_path"  the same step without the two CNNs (features / regularised volumes are leaves): volume build + render + backward
    #   "finetune"  per-scene fine-tune: the volumes are the parameters (no CNN, no volume build)
    # Run after the headline's timed region and kernel table; reported beside the headline, never as `value`.
    train = val_item = None
    if secondary:
        state.clear()
        torch.cuda.empty_cache()
        try:
            from scripts.train_step_bench import measure
            train = {"workload": "BASELINE config[2]: DTU-shaped training step, 5 views 480x640, volume_dims [256, 128, 64], 512 rays + 2048 pseudo "
                                 "points, the reference's Loss (gens_amd.losses.Loss, shipped weights) + backward + torch.optim.Adam(model.get_optim_params(...)) "
                                 "as runner.py:96-97 builds it; 30 timed steps after 5 warm-up each, the loss read back every step as runner.py does; "
                                 "ms_per_step = the MEDIAN step (mean_ms_per_step beside it)",
                     "note": ("secondary figures; not the headline.  hot_path / finetune / finetune_conf / full: the loop of runner.py:157-166 / 300-308 AS "
                              "WRITTEN -- model(...), loss, zero_grad, backward, optimizer.step, loss read back -- with no graph object in the caller: behind "
                              "GenS.forward / ImplicitSurface.forward the step's forward and backward replay from two HIP graphs captured after two eager "
                              "calls (gens_amd.graph.AutoGraph).  *_graph: the caller captures the WHOLE step incl. loss and optimiser itself "
                              "(gens_amd.graph.GraphedStep): the lower bound.  *_nograph: every call eager (GENS_AUTO_GRAPH=0; rounds 1 - 4's eager "
                              "figures: ~140 - 1 070 launches per step enqueued from Python, host-paced)")}
            runs = (("hot_path", []), ("finetune", ["--finetune"]), ("finetune_conf", ["--finetune", "--conf-shape"]), ("full", ["--full"]),
                    ("hot_path_graph", ["--graph"]), ("finetune_graph", ["--finetune", "--graph"]),
                    ("finetune_conf_graph", ["--finetune", "--conf-shape", "--graph"]), ("full_graph", ["--full", "--graph"]),
                    ("finetune_nograph", ["--finetune", "--no-auto"]), ("full_nograph", ["--full", "--no-auto"]),
                    ("finetune_foreach_adam", ["--finetune", "--foreach-adam"]))
            for key, flags in runs:
                try:
                    ms, label, kt = measure(flags + ["--steps", "30", "--warm", "5"], quiet=True, kernels=key in ("hot_path", "finetune", "finetune_conf", "full"))
                except Exception as e:                                 # (a secondary of the secondaries: report, do not lose the others)
                    train[key] = {"error": f"{type(e).__name__}: {e}"}
                    continue
                from scripts.train_step_bench import _measure as _m
                stats = dict(getattr(_m, "stats", {}))
                auto_stats = stats.pop("auto_graph", None)
                med = stats.get("median_ms", ms)                      # the median; the mean of an eager step carries the host's hiccups (both are reported)
                train[key] = {"what": label, "ms_per_step": round(med, 2), "mean_ms_per_step": round(ms, 2), "ms_per_step_stats": stats,
                              "ray_samples_per_s": round(512 * 128 / med * 1e3, 1)}
                if auto_stats is not None:
                    train[key]["auto_graph"] = auto_stats             # eager warm-up calls / captures / replays behind forward()
                if kt:
                    train[key]["launches_per_step_c_abi"] = sum(k["launches"] for k in kt.values())
                    train[key]["hip_kernels"] = kernel_rows(kt, 8)
                    train[key]["roofline"] = kernel_roofline(kt)
                torch.cuda.empty_cache()
            train["finetune_conf"].setdefault("workload", None)
            train["finetune_conf"]["workload"] = ("confs/gens_finetune.conf as shipped (BASELINE config[4] on one GPU): img_hw 1152 x 1600, num_views 3, "
                                                  "volume_dims 256/128/64/32/16 as parameters, 512 rays + 2048 pseudo points")
            train["ms_per_step"] = train["full"].get("ms_per_step")
            if "error" not in train["finetune_foreach_adam"]:
                train["finetune_foreach_adam"]["note"] = ("the fine-tune loop with GENS_FUSED_ADAM=0: torch's multi-tensor Adam makes ~10 passes over the 307 MB of "
                                                          "volumes where the fused update get_optim_params asks for makes one")
            # the boundary promise in one place: the unchanged loop against the caller-captured whole step
            ratios = {}
            for key in ("hot_path", "finetune", "finetune_conf", "full"):
                a, b = train.get(key, {}), train.get(key + "_graph", {})
                if "ms_per_step" in a and "ms_per_step" in b:
                    ratios[key] = {"unchanged_loop_over_whole_step_graph": round(a["ms_per_step"] / b["ms_per_step"], 3),
                                   "p90_over_median": round(a["ms_per_step_stats"]["p90_ms"] / a["ms_per_step"], 3)}
            train["unchanged_loop_vs_graph"] = ratios
        except Exception as e:                                             # never let a secondary figure take the headline down
            train = {"error": f"{type(e).__name__}: {e}"}
        # secondary figure: one whole `--mode val` item (volume build, 512^3 SDF lattice, marching cubes on the device, 480x640 render)
        try:
            from scripts.val_full_bench import measure as val_measure
            val_item = {k: (round(v, 2) if isinstance(v, float) else v) for k, v in val_measure(repeats=SECONDARY_STEPS + 1).items()}
            val_item["workload"] = "BASELINE config[1] as runner.py --mode val runs it: K1 + 512^3 lattice + iso-surface + 307 200-ray render"
            half = val_measure(repeats=SECONDARY_STEPS + 1, sdf_precision="f16x2")      # the same item in the opt-in split-half arithmetic
            val_item["split_half"] = {k: round(half[k], 2) if isinstance(half[k], float) else half[k]
                                      for k in ("lattice_ms", "render_ms", "total_ms", "total_ms_stats", "vertices", "triangles")}
        except Exception as e:
            val_item = {"error": f"{type(e).__name__}: {e}"}
    # The measured configuration IS the shipped one: the headline above sets nothing on the model (ray chunk and jitter head start are
    # ImplicitSurface.validate's own defaults).  Beside it, the same scene as ONE validation item through the public boundary --
    # GenS(has_vol).forward("val", ipts), runner.py:215 -- with the item's mesh time taken out: within 2 % of the headline step.
    default_path = None
    if secondary and args.sdf_precision == "f32" and not args.chunk:
        try:
            from scripts.val_full_bench import measure_default_path
            default_path = measure_default_path(repeats=SECONDARY_STEPS + 5, dims=tuple(args.dims))      # (the first three items are left out)
            headline_ms = elapsed / args.steps * 1e3
            default_path.update({
                "ms_per_step": default_path["render_ms"], "value": round(n_rays * n_final / default_path["render_ms"] * 1e3, 1), "unit": "ray-samples/s",
                "ratio_to_headline_ms": round(default_path["render_ms"] / headline_ms, 4),
                "what": "GenS(has_vol).forward('val', ipts) as runner.py:215 calls it, nothing set on the model from outside; item_ms = the whole item "
                        "incl. the mesh the reference's validate always extracts (512^3 lattice + marching cubes = geometry_ms, wall time taken inside "
                        "validate), render_ms = the image's part of the same call, timed inside validate from the end of the mesh's read-back to the image "
                        "on the host = the headline's step without K1 (~0.2 ms; the volumes of a has_vol model are parameters), rest_ms = what "
                        "GenS.forward does around validate on the host (scene set-up, the mesh into world space, the outputs); render_alone_ms = "
                        "validate() called directly with the geometry off"})
        except Exception as e:
            default_path = {"error": f"{type(e).__name__}: {e}"}

    line = {
        "metric": "SDF ray-samples/sec at 480x640, 5-view, 3-scale volumes", "value": value, "unit": "ray-samples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "ms_per_step_stats": percentiles(step_ms), "ms_per_step_by_rank": {"min": round(min(rank_ms), 3), "max": round(max(rank_ms), 3)},
        "clocks": clocks.summary(),
        "higher_is_better": True, "scaling": "strong" if by_rays else "weak", "vs_baseline": None,
        "dtype": "f32" if args.sdf_precision == "f32" else "f32 (SDF-MLP operands as split f16 hi+lo pairs, f32 accumulate)", "data": "synthetic",
        "config": {"workload": "BASELINE config[1]: 5-view 480x640, volume_dims=%s, inference of %d rays x %d samples per scene "
                               "(K1 volume build + hierarchical sampling + SDF/blend MLPs + compositing); one scene per GPU"
                               % (args.dims, n_rays, n_final),
                   "rays_per_step_per_gpu": n_rays, "samples_per_ray": n_final, "views": args.views, "volume_dims": args.dims,
                   "ray_chunk": headline_chunk, "ray_chunk_set_by": "--chunk" if args.chunk else "ImplicitSurface.validate (default)", "cnn": "out of scope (synthetic feature pyramid and regularised volumes)",
                   "parallelism": ("ONE scene, contiguous ray ranges across ranks, K1 replicated, all_gather of rendered buffers in the timed region"
                                   if by_rays else "scenes sharded across ranks, all_gather of rendered buffers") if world > 1 else "single GPU"},
        "value_note": (None if world == 1 else "ray-sharded (strong scaling): ONE scene's rays split across the ranks" if by_rays else
                       "weak scaling: `value` is %d INDEPENDENT scenes, one per GPU (what the reference's DistributedSampler does, datasets/__init__.py:33) -- "
                       "N x one GPU by construction; the figure that answers north_star's '>= 6 x ray-throughput at 8 GPUs' is `strong_scaling` "
                       "(one scene, its rays split across the ranks, gather inside the timed region)" % world),
        "strong_scaling": (None if ray_sharded is None or "error" in ray_sharded else
                           {k: ray_sharded[k] for k in ("value", "unit", "ms_per_step", "ms_per_step_by_rank", "n_gpus", "rays_per_rank", "ray_chunk")}),
        "k1_in_step": k1_in_step, "default_path": default_path,
        "roofline": roofline, "cpu_baseline": cpu, "split_half_sdf": split, "levels5": levels5, "views3": views3, "train_step": train, "val_item": val_item,
        "ray_sharded": ray_sharded, "ray_sharded_projection": projection,
        "hip_kernels": table,
    }
    os.write(real_stdout, (json.dumps(line) + "\n").encode())
    finish()


def kernel_rows(table, top):
    """{entry: {launches, ms, bytes, flops}} of one step -> the `top` slowest as rows with their algorithmic rates."""
    rows = {}
    for name, k in sorted(table.items(), key=lambda kv: -kv[1]["ms"])[:top]:
        rows[name] = {"launches": k["launches"], "ms_per_step": round(k["ms"], 3)}
        if k["ms"] > 0 and k["bytes"]:
            rows[name]["algo_GBs"] = round(k["bytes"] / 1e9 / (k["ms"] / 1e3), 1)
        if k["ms"] > 0 and k.get("flops"):
            rows[name]["TFLOPs"] = round(k["flops"] / 1e12 / (k["ms"] / 1e3), 1)
        if name == "gens_blend_train_bwd" and "TFLOPs" in rows[name]:
            from gens_amd.ops.base import kernels
            ratio = getattr(kernels, "blend_bwd_round5_count_ratio", None)
            if ratio:
                rows[name]["TFLOPs_by_round5_count"] = round(rows[name]["TFLOPs"] * ratio, 1)
                rows[name]["flops_note"] = ("TFLOPs counts the forward again + the reverse chain + [dW | db] = (2 F + 2 S sum m (k + 1)) per point; rounds 4 - 5 "
                                            "counted 3 F for the chain (a third too much): TFLOPs_by_round5_count is the figure comparable with their 32 - 34")
    return rows


def kernel_roofline(table):
    """Roofline object of the slowest C-ABI kernel of a step that carries an algorithmic count (HIP events on the launch stream)."""
    cand = [(n, k) for n, k in table.items() if k["ms"] > 0 and (k["bytes"] or k.get("flops"))]
    if not cand:
        return None
    name, k = max(cand, key=lambda kv: kv[1]["ms"])
    if k.get("flops"):
        a = k["flops"] / 1e12 / (k["ms"] / 1e3)
        return {"kernel": name, "bound": "mfma", "achieved": round(a, 1), "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(a / F32_MFMA_PEAK_TFLOPS, 4), "avg_launch_us": round(k["ms"] * 1e3 / k["launches"], 1)}
    a = k["bytes"] / 1e9 / (k["ms"] / 1e3)
    return {"kernel": name, "bound": "hbm", "achieved": round(a, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(a / HBM_PEAK_GBS, 4),
            "avg_launch_us": round(k["ms"] * 1e3 / k["launches"], 1)}


def ray_sharded_variant(args, dev, dist, surf, volume, n_final, sync, projection=None):
    """ONE scene (seed 0 on every rank), its 307 200 rays split into contiguous ranges across the ranks, K1 replicated, the rendered (P, 8)
    buffers all-gathered over RCCL inside the timed region: whole-job ray-samples/s with the work FIXED as N grows (strong scaling).

    projection=(rank, world), one GPU, no process group: the time of THAT rank's share rendered alone (Shard.single: K1 + its ray range, no
    gather) -- a labelled 1-GPU projection of the ray-sharded step, not a scaling measurement."""
    from gens_amd import synthetic
    from gens_amd.distributed import Shard
    from gens_amd.models.modules.implicit_surface import Scene
    sc = synthetic.make_scene(nv=args.views, h=480, w=640, n_levels=5, seed=0)
    imgs, intrs, c2ws = sc["imgs"].to(dev), sc["intrs"].to(dev), sc["c2ws"].to(dev)
    feats = [f.to(dev) for f in sc["features"]]
    near, far = sc["near"].to(dev), sc["far"].to(dev)
    vols = [v.to(dev) for v in synthetic.make_volumes(args.dims, seed=100)]
    rays_o, rays_d = synthetic.make_rays(sc["intrs"], sc["c2ws"], 480, 640)
    rays_o, rays_d = rays_o[:args.rays].to(dev), rays_d[:args.rays].to(dev)
    n_rays = rays_o.shape[0]
    sink = {}
    shard = Shard() if projection is None else Shard.single(projection[0], projection[1], sink)
    r0, r1 = shard.rays(n_rays)
    ray_chunk = balanced_chunk(r1 - r0, args.chunk) if args.chunk else balanced_chunk(r1 - r0)      # what validate() chooses for this rank's range

    def step():
        sink.clear()
        with torch.no_grad():
            _, masks = volume.agg_mean_var(feats, intrs, c2ws)
            scene = Scene(vols, masks, imgs, feats, feats, intrs, c2ws)
            surf.validate(rays_o, rays_d, near, far, vols, masks, imgs, feats, feats, intrs, c2ws, None, None, (1, n_rays), extract_geometry=False,
                          scene=scene, shard=shard)
    torch.manual_seed(4321)                    # the same CPU generator state on every rank: identical jitter for every ray
    step()
    sync()
    steps = SECONDARY_STEPS
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    dt = (time.perf_counter() - t0) / steps
    if projection is not None:
        return {"ms": round(dt * 1e3, 2), "steps": steps, "rank": projection[0], "world": projection[1], "rays": r1 - r0, "ray_chunk": ray_chunk,
                "what": "ONE GPU renders rank %d's share of a %d-way ray-sharded image alone: K1 (replicated on every rank) + rays [%d, %d) of %d, "
                        "no gather; a projection of the ray-sharded step, NOT a multi-GPU measurement" % (projection[0], projection[1], r0, r1, n_rays)}
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    every = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(every, t)                                   # each rank's own time: min / max beside the figure (the slowest rank is the step)
    per_rank = [float(x) * 1e3 for x in every]
    dt = max(per_rank) / 1e3
    return {"scaling": "strong", "value": n_rays * n_final / dt, "unit": "ray-samples/s", "ms_per_step": round(dt * 1e3, 2), "steps": steps,
            "ms_per_step_by_rank": {"min": round(min(per_rank), 3), "max": round(max(per_rank), 3)},
            "n_gpus": dist.get_world_size(), "ray_chunk": ray_chunk, "rays_per_rank": r1 - r0,
            "workload": "one scene, %d rays split across the ranks, all_gather of the (P, 8) buffers in the timed region" % n_rays,
            "note": "secondary figure of the same run; `python bench.py --gpus N --shard rays` reports it as the headline"}


def five_level_variant(args, dev, sc, feats, imgs, intrs, c2ws, near, far, rays_o, rays_d, n_final, kernels=False):
    """The headline step with the shipped five-level pyramid (volume_dims 256 / 128 / 64 / 32 / 16: sdf_mlp_k<100>): SECONDARY_STEPS timed steps.
    kernels: add the C-ABI kernel table of one extra, untimed step (HIP events per launch)."""
    from gens_amd import synthetic
    from gens_amd.models.modules.implicit_surface import Scene
    dims = [256, 128, 64, 32, 16]
    surf, volume = build_model(dims, dev)
    if args.chunk:
        surf.val_chunk = args.chunk
    vols = [v.to(dev) for v in synthetic.make_volumes(dims, seed=100)]
    n_rays = rays_o.shape[0]

    def step():
        with torch.no_grad():
            _, masks = volume.agg_mean_var(feats, intrs, c2ws)
            scene = Scene(vols, masks, imgs, feats, feats, intrs, c2ws)
            surf.validate(rays_o, rays_d, near, far, vols, masks, imgs, feats, feats, intrs, c2ws, None, None, (1, n_rays), extract_geometry=False,
                          scene=scene)
    step()
    torch.cuda.synchronize()
    import gc
    gc.collect()
    dt, ms = run_timed(step)
    res = {"volume_dims": dims, "value": n_rays * n_final / dt, "unit": "ray-samples/s", "ms_per_step": round(dt * 1e3, 2), "steps": SECONDARY_STEPS,
           "ms_per_step_stats": percentiles(ms), "note": "the shipped level count of confs/gens.conf; BASELINE's metric is quoted on three levels, so this is not the headline"}
    if not kernels:      # the same steps in the opt-in split-half arithmetic (gens_sdf_value_f16 / gens_sdf_grad_f16 at five levels)
        surf.sdf_precision = "f16x2"
        step()
        torch.cuda.synchronize()
        res["split_half_ms_per_step"] = round(run_timed(step)[0] * 1e3, 2)
        surf.sdf_precision = "f32"
    if kernels:
        from gens_amd import lib as L
        L.profile_begin()
        step()
        torch.cuda.synchronize()
        res["hip_kernels"] = kernel_rows(L.profile_end(), 4)
    surf.join_speculation()
    return res


def cpu_baseline(args, surf, sc, vols, masks, n_final):
    """The CPU oracle (restatement of the reference's render(), validated against its goldens) on a bounded sample."""
    from oracle import render_oracle as R
    n = args.cpu_rays
    threads = torch.get_num_threads()
    sd = {k: v.detach().cpu() for k, v in surf.state_dict().items()}
    g = torch.Generator().manual_seed(5)
    idx = torch.randint(0, 480 * 640, (n,), generator=g)
    from gens_amd import synthetic
    ro, rd = synthetic.make_rays(sc["intrs"], sc["c2ws"], 480, 640)
    ro, rd = ro[idx], rd[idx]
    cvols = [v.cpu() for v in vols]
    cmasks = [m.cpu() for m in masks]
    t_rand, pts_rand = torch.rand(n, 1, generator=g), torch.rand(1024, 3, generator=g) * 2 - 1
    t0 = time.perf_counter()
    R.render(sd, ro, rd, sc["near"], sc["far"], cvols, cmasks, sc["imgs"], sc["features"], sc["features"], sc["intrs"], sc["c2ws"], 1.0,
             None, t_rand, pts_rand)
    dt = time.perf_counter() - t0
    return {"value": n * n_final / dt, "unit": "ray-samples/s", "cores": threads, "kind": "port",
            "sample": "%d random rays of the same scene x %d samples through oracle.render_oracle.render, %.1f s.  The oracle runs the FULL "
                      "render_core as the reference's validate does (second-order `smooth` terms, random-point SDF, TV, surface-point gradient "
                      "and patch warp, all of which validate then discards, implicit_surface.py:446-453); the GPU path is the lean validate "
                      "path that skips exactly those discarded quantities, so the ratio overstates the kernels' advantage" % (n, n_final, dt)}


if __name__ == "__main__":
    main()
