#!/usr/bin/env python3
"""Loops ONE kernel of the step for a few seconds (for clock / power sampling from outside):
    python scripts/experiments/rhs_loop.py rhs|visc|pres [seconds] [--igc N]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from microhh_amd.model import HotPath   # noqa: E402
what = sys.argv[1] if len(sys.argv) > 1 else "rhs"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
igc = int(sys.argv[sys.argv.index("--igc") + 1]) if "--igc" in sys.argv else None
hp = HotPath("drycblles", 512, 512, 512, igc=igc)
hp.cyclic_prognostic(); hp.exec_viscosity(); hp.sync()
fn = {"rhs": hp.rhs, "visc": hp.exec_viscosity, "pres": hp.pres}[what]
import torch
t0 = time.time(); n = 0
while time.time() - t0 < secs:
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        fn()
    b.record(); hp.sync(); n += 50
    last = a.elapsed_time(b) / 50
print("%s: %d launches, last batch %.3f ms per launch" % (what, n, last))
hp.close()
