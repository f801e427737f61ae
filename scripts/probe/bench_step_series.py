#!/usr/bin/env python3
"""Per-step wall time of bench.py's step over the first steps of a process (each step synchronised), with the allocator's counters:
why do 3 timed steps after 1 warm-up measure ~290 ms where 10 after 3 measure ~277?"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

# run bench.main() with a patched perf_counter-free hook: simplest is to re-create its step through the same helpers
sys.argv = [sys.argv[0], "--steps", "1", "--warmup", "0", "--cpu-rays", "0", "--no-kernel-timing"]
orig_sync = torch.cuda.synchronize
times = []


def main():
    # bench.main() builds everything and runs warm-up 0 + 1 timed step; we wrap ImplicitSurface.validate to time every image
    from gens_amd.models.modules import implicit_surface as isurf
    orig = isurf.ImplicitSurface.validate

    def timed(self, *a, **k):
        orig_sync()
        t0 = time.perf_counter()
        out = orig(self, *a, **k)
        orig_sync()
        st = torch.cuda.memory_stats()
        times.append(((time.perf_counter() - t0) * 1e3, st["num_alloc_retries"], st["segment.all.allocated"], st["reserved_bytes.all.current"] / 1e9))
        return out
    isurf.ImplicitSurface.validate = timed
    from gens_amd.models.modules import volume as vmod
    orig_agg = vmod.Volume.agg_mean_var

    def timed_agg(self, *a, **k):
        orig_sync()
        t0 = time.perf_counter()
        out = orig_agg(self, *a, **k)
        orig_sync()
        st = torch.cuda.memory_stats()
        print(f"   volume build {(time.perf_counter() - t0) * 1e3:7.1f} ms   segments {st['segment.all.allocated']:4d}   reserved {st['reserved_bytes.all.current'] / 1e9:5.2f} GB", file=sys.stderr)
        return out
    vmod.Volume.agg_mean_var = timed_agg
    sys.argv[2] = "12"
    bench.main()
    for i, (ms, retries, segs, gb) in enumerate(times):
        print(f"image {i:2d}: validate {ms:7.1f} ms   segments allocated so far {segs:5d}   reserved {gb:6.2f} GB", file=sys.stderr)


if __name__ == "__main__":
    main()
