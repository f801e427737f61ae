#!/usr/bin/env python3
"""After a fine-tune measurement returns: are its captured graphs garbage, and if not, who holds them?"""
import gc
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scripts.train_step_bench import measure  # noqa: E402


def graphs():
    return [o for o in gc.get_objects() if isinstance(o, torch.cuda.CUDAGraph)]


measure(["--finetune", "--steps", "5", "--warm", "5"], quiet=True, kernels=True)
print("after measure():", len(graphs()))
print("collected", gc.collect(), "->", len(graphs()))
gs = graphs()
if gs:
    seen = set()
    frontier = [gs[0]]
    for depth in range(7):
        nxt = []
        for o in frontier:
            for r in gc.get_referrers(o):
                if id(r) in seen or r is frontier or r is gs or r is nxt:
                    continue
                seen.add(id(r))
                desc = type(r).__name__
                if isinstance(r, dict):
                    desc += " keys=" + ",".join(str(k)[:20] for k in list(r)[:6])
                elif hasattr(r, "__name__"):
                    desc += " " + str(getattr(r, "__name__", ""))
                print("  " * depth, "<-", desc[:150])
                nxt.append(r)
        frontier = nxt[:6]
