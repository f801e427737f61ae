"""CPU tests of host-side logic that needs no GPU: the reference-order jitter stream, the config stand-in."""
import torch

from gens_amd.config import gens_model_conf
from gens_amd.models.modules import implicit_surface as M


def _reference_order(n, chunk=256):
    out = []
    for start in range(0, n, chunk):
        out.append(torch.rand([min(chunk, n - start), 1]))      # render(), implicit_surface.py:362
        torch.rand([1024, 3])                                   # render_core(), implicit_surface.py:256
    return torch.cat(out, 0)


def test_jitter_matches_the_reference_draw_order_for_any_chunking():
    for n in (1, 24, 256, 257, 1000, 70000):
        torch.manual_seed(7)
        ref = _reference_order(n)
        after_ref = torch.rand(4)
        torch.manual_seed(7)
        one = M.reference_jitter(n)
        after_one = torch.rand(4)
        torch.manual_seed(7)
        js = M.JitterStream(n, group=8192)
        parts = torch.cat([js.slice(s, min(s + 5000, n)) for s in range(0, n, 5000)])
        js.join()
        after_stream = torch.rand(4)
        assert torch.equal(ref, one) and torch.equal(ref, parts)
        assert torch.equal(after_ref, after_one) and torch.equal(after_ref, after_stream)   # generator left in the same state


def test_jitter_head_start_is_used_only_if_the_generator_has_not_moved(monkeypatch):
    """ImplicitSurface._speculate_jitter / _take_speculated_jitter (what validate() does between two images), without a device."""
    surf = M.ImplicitSurface(gens_model_conf(volume_dims=(8, 4, 2))["implicit_surface"])
    monkeypatch.setattr(surf, "_pinned", lambda n, cols=8, slot="": torch.empty(n, cols))
    n = 700
    # used: the draws equal the reference's from that state, and end_state() is the state behind them
    torch.manual_seed(5)
    start = torch.get_rng_state()
    surf._speculate_jitter(n)
    assert torch.equal(torch.get_rng_state(), start)                 # the default generator did not move
    js = surf._take_speculated_jitter(n)
    assert js is not None and surf._jitter_ahead is None
    got = js.slice(0, n).clone()
    js.join()
    want = M.reference_jitter(n)
    assert torch.equal(got, want) and torch.equal(js.end_state(), torch.get_rng_state())
    # void: somebody drew in between, or another ray count is asked for
    torch.manual_seed(5)
    surf._speculate_jitter(n)
    torch.rand(3)
    state = torch.get_rng_state()
    assert surf._take_speculated_jitter(n) is None and torch.equal(torch.get_rng_state(), state)
    surf._speculate_jitter(n)
    assert surf._take_speculated_jitter(n + 1) is None and torch.equal(torch.get_rng_state(), state)
    # asked twice from the same state: one thread, not two
    surf._speculate_jitter(n)
    first = surf._jitter_ahead[2]
    surf.prefetch_jitter(n)
    assert surf._jitter_ahead[2] is first
    surf.join_speculation()
    assert surf.val_chunk is None and surf.val_chunk_for(307200) == 30720 and surf.val_chunk_for(38400) == 19200
    surf.val_chunk = 512
    assert surf.val_chunk_for(307200) == 512


def test_config_stand_in_behaves_like_a_config_tree():
    c = gens_model_conf(volume_dims=(256, 128, 64))
    assert c.get_list("volume.volume_dims") == [256, 128, 64]
    assert c["implicit_surface"].get_int("render.n_samples") == 64
    assert c.get_bool("has_vol", default=False) is False
    assert c["implicit_surface"]["sdf_network"]["feat_channels"] == 12
    assert dict(**c["implicit_surface"]["color_network"]) == {"d_feature": 20}


def test_bench_command_line_contract():
    """bench.py's flags as the driver uses them (`--gpus N --steps K --warmup W`), its defaults, and that the opt-in extras stay off."""
    import importlib
    import sys
    root = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    bench = importlib.import_module("bench")
    saved = sys.argv
    try:
        sys.argv = ["bench.py"]
        a = bench.parse()
        assert (a.gpus, a.steps, a.warmup, a.rays, a.dims, a.views) == (1, 10, 3, 480 * 640, [256, 128, 64], 5)
        assert a.sdf_precision == "f32" and not a.train_step and not a.no_kernel_timing
        sys.argv = ["bench.py", "--gpus", "8", "--steps", "5", "--warmup", "2"]
        a = bench.parse()
        assert (a.gpus, a.steps, a.warmup) == (8, 5, 2)
    finally:
        sys.argv = saved


# ---------------------------------------------------------------------------------------------------------------------------------
# bench.py --gpus N launches its own rank processes (the reference: scripts/run.sh:3 + utils/distribute.py:66-88)
# ---------------------------------------------------------------------------------------------------------------------------------
def _run_py(code, env=None, timeout=120):
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ if env is None else env)
    e["PYTHONPATH"] = root + os.pathsep + e.get("PYTHONPATH", "")
    return subprocess.run([sys.executable, "-c", code], env=e, cwd=root, capture_output=True, text=True, timeout=timeout)


def test_bench_gpus_n_starts_n_ranks_and_relays_rank_zero_without_loading_torch():
    """The parent of `bench.py --gpus N` starts N children with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, relays rank 0's stdout (the
    one JSON line) and sends the others' to stderr -- and never loads torch, so it cannot have initialised HIP."""
    code = r'''
import json, sys
import bench
stub = [sys.executable, "-c", "import os, json; print(json.dumps({k: os.environ[k] for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}))"]
rc = bench.launch_ranks(3, ["--gpus", "3"], child=stub)
assert rc == 0, rc
assert "torch" not in sys.modules, "the launching parent loaded torch"
'''
    r = _run_py(code)
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                              # ONE line on stdout: rank 0's
    import json
    zero = json.loads(lines[0])
    assert zero["RANK"] == "0" and zero["LOCAL_RANK"] == "0" and zero["WORLD_SIZE"] == "3" and zero["MASTER_ADDR"] == "127.0.0.1"
    others = sorted(json.loads(ln)["RANK"] for ln in r.stderr.splitlines() if ln.startswith("{"))
    assert others == ["1", "2"]


def test_bench_launcher_stops_the_other_ranks_when_one_dies():
    """A rank that exits non-zero ends the run: the others (here: one that would sleep for ten minutes) are terminated and the parent's exit
    status is the failing rank's; nothing of rank 0's output reaches stdout."""
    import time
    code = r'''
import sys
import bench
stub = [sys.executable, "-c", "import os, sys, time; r = int(os.environ['RANK']); print('line of rank', r); sys.stdout.flush(); sys.exit(7) if r == 1 else time.sleep(600)"]
sys.exit(bench.launch_ranks(2, [], child=stub))
'''
    t0 = time.time()
    r = _run_py(code)
    assert r.returncode == 7, (r.returncode, r.stderr)
    assert time.time() - t0 < 60
    assert r.stdout.strip() == ""
    assert "rank 1 exited with status 7" in r.stderr


def test_bench_launcher_interrupted_by_sigterm_leaves_no_rank_behind(tmp_path):
    """A harness timeout SIGTERMs the PARENT: the rank processes (each writes its pid, then would sleep for ten minutes -- in a real run: wait
    in an RCCL collective holding its GPU) must be gone when the parent exits, and the rendezvous directory with them."""
    import os
    import signal
    import subprocess
    import sys
    import time
    code = r'''
import sys
import bench
stub = [sys.executable, "-c", "import os, time; open(os.path.join(%r, 'pid%%s' %% os.environ['RANK']), 'w').write(str(os.getpid())); time.sleep(600)"]
sys.exit(bench.launch_ranks(3, [], child=stub))
''' % str(tmp_path)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    parent = subprocess.Popen([sys.executable, "-c", code], cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    deadline = time.time() + 60
    while time.time() < deadline and len(list(tmp_path.glob("pid*"))) < 3:
        time.sleep(0.05)
    pids = [int(f.read_text()) for f in tmp_path.glob("pid*") if f.read_text()]
    assert len(pids) == 3
    parent.send_signal(signal.SIGTERM)
    out, err = parent.communicate(timeout=60)
    assert parent.returncode == 143, (parent.returncode, err)
    assert b"launcher interrupted" in err
    for pid in pids:
        for _ in range(100):                                      # (reaped by the parent's wait(); give the kernel a moment)
            try:
                os.kill(pid, 0)
            except ProcessLookupError:
                break
            time.sleep(0.05)
        else:
            raise AssertionError("rank process %d survived its launcher" % pid)


def test_bench_main_becomes_the_launcher_only_without_a_launcher_around_it():
    """--gpus 2 with no WORLD_SIZE: main() hands over to launch_ranks before torch is imported.  Under a launcher (WORLD_SIZE set) --gpus must
    equal the world size or the run aborts."""
    code = r'''
import sys
import bench
calls = []
bench.launch_ranks = lambda n, argv, **kw: calls.append((n, list(argv), "torch" in sys.modules)) or 0
sys.argv = ["bench.py", "--gpus", "2", "--steps", "2"]
try:
    bench.main()
except SystemExit as e:
    assert e.code == 0, e.code
assert calls == [(2, ["--gpus", "2", "--steps", "2"], False)], calls
'''
    import os
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = _run_py(code, env=env)
    assert r.returncode == 0, r.stderr
    code = r'''
import sys
import bench
sys.argv = ["bench.py", "--gpus", "8"]
try:
    bench.main()
except SystemExit as e:
    assert "WORLD_SIZE=2" in str(e.code) and "--gpus 8" in str(e.code), e.code
    assert "torch" not in sys.modules
else:
    raise AssertionError("a world size that contradicts --gpus was accepted")
'''
    r = _run_py(code, env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"))
    assert r.returncode == 0, r.stderr


def test_bench_balanced_chunks_cover_a_shard_without_a_short_tail():
    import importlib
    bench = importlib.import_module("bench")
    assert bench.balanced_chunk(38400, 32768) == 19200            # an eighth of 480 x 640: 2 x 19 200, not 32 768 + 5 632
    assert bench.balanced_chunk(153600, 32768) == 30720
    assert bench.balanced_chunk(32768, 32768) == 32768
    for n in (1, 255, 256, 257, 5000, 38400, 76800, 307200):
        c = bench.balanced_chunk(n, 32768)
        assert c % 256 == 0 and c <= 32768
        lens = [min(c, n - s) for s in range(0, n, c)]
        assert sum(lens) == n and len(lens) == -(-n // 32768) or n <= 32768
