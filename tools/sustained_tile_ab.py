#!/usr/bin/env python
"""Does the autotuner's 4-launch timing (boost clock) pick the tiles that are fastest under SUSTAINED load (power-limited clock)?  Runs bench.py's
headline with the tap kernels' candidate list restricted by FGT_TAPS_ONLY (comma-separated tile names; layers every listed tile declines fall back to
the static tile) and a fresh tuning table.    FGT_TAPS_ONLY=256x256it,256x128it python tools/sustained_tile_ab.py --steps 20 --warmup 5 ..."""
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fgt_amd import ops  # noqa: E402

only = os.environ.get("FGT_TAPS_ONLY")
if only:
    ops.TAPS_CANDIDATES = tuple(t for t in only.split(",") if t)
sys.argv = ["bench.py"] + sys.argv[1:]
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), run_name="__main__")
