#!/usr/bin/env python3
"""Host side of a training step: cProfile over the timed steps of scripts/train_step_bench.py (same flags), top functions by own and cumulative time."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import train_step_bench  # noqa: E402

prof = cProfile.Profile()
prof.enable()
ms, label = train_step_bench.measure(sys.argv[1:], quiet=True)
prof.disable()
print(f"{label}: {ms:.2f} ms/step under cProfile")
for key in ("tottime", "cumtime"):
    st = pstats.Stats(prof)
    st.sort_stats(key).print_stats(28)
