#!/usr/bin/env python3
"""Where a level of the fused 2i5+smag2 marching kernel spends its cycles, per wave: a probe build of k_march.hip
(-DMHH_MARCH_STAMP, microhh_amd.build.build_variant_of("stamp", ...)) sums s_memtime differences over the phases of every level
and writes them per wave at the end of the kernel.
    MHH_LIB=$PWD/microhh_amd/variants/libmhh_hip_stamp.so python scripts/experiments/march_stamps.py [itot jtot ktot] [--igc N]
Phases: issue (LDS-DMA copies, column and tendency loads of the next level) | compute (LDS reads, arithmetic, stores) |
vmem (s_waitcnt vmcnt(0) in front of the barrier) | barrier | loop (window rotation, loop control)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from microhh_amd.model import HotPath   # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
igc = int(sys.argv[sys.argv.index("--igc") + 1]) if "--igc" in sys.argv else None
n = [int(a) for a in args[:3]] if len(args) >= 3 else [512, 512, 512]
hp = HotPath("drycblles", *n, igc=igc)
hp.cyclic_prognostic(); hp.exec_viscosity()
for _ in range(3):
    hp.rhs()
hp.sync()
lib = hp.lib
lib.mhh_march_stamps.restype = C.c_longlong
lib.mhh_march_stamps.argtypes = [C.c_void_p, C.c_longlong]
buf = np.zeros(64 << 20, dtype=np.uint64)
got = lib.mhh_march_stamps(buf.ctypes.data, buf.size)
assert got > 0, got
s = buf[:got].reshape(-1, 8).astype(np.float64)
s = s[s[:, 5] > 0]
lev = s[:, 5]
names = ("issue", "compute", "vmem wait", "barrier", "loop")
tot = s[:, :5].sum(axis=1)
print("waves %d, levels per wave %.1f, cycles per level %.0f (s_memtime ticks at the shader clock's reference: 100 MHz -> see below)" % (len(s), lev.mean(), (tot / lev).mean()))
for i, nm in enumerate(names):
    per = s[:, i] / lev
    print("  %-10s mean %8.1f  p10 %8.1f  p50 %8.1f  p90 %8.1f   share %.3f" % (nm, per.mean(), *np.percentile(per, [10, 50, 90]), s[:, i].sum() / tot.sum()))
hp.close()
