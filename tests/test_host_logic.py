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
