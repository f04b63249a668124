"""Diagnostic: run the three modes of the 2i5+smag2 marching kernel one by one on a small grid, synchronising after each."""
import ctypes as C
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import backends as B
import common as cm
from microhh_amd import capi

be = B.get("hip")
shape = tuple(int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (16, 12, 10)
for dtype in (np.float64, np.float32):
    for rho in ("one", "random"):
        for mode in ("advec", "diff", "fused"):
            g = cm.grid_2nd(*shape, gc=(3, 3, 1), dtype=dtype)
            c = cm.Case(g, nscalars=1, rho=rho)
            p = capi.MhhDiffParams(); p.cs = 0.23; p.tPr = 1./3.; p.surface_model = 1
            d = B.DevCase(be, c); f = d.fields()
            print("run", dtype.__name__, rho, mode, flush=True)
            if mode == "advec":
                rc = be.lib.mhh_advec_exec(d.G, cm.ADVEC_2I5, C.byref(f), be.stream)
            elif mode == "diff":
                rc = be.lib.mhh_diff_exec(d.G, cm.DIFF_SMAG2, C.byref(f), C.byref(p), be.stream)
            else:
                rc = be.lib.mhh_rhs_exec(d.G, cm.ADVEC_2I5, cm.DIFF_SMAG2, C.byref(f), C.byref(p), be.stream)
            torch.cuda.synchronize()
            print("  rc", rc, "sum ut", float(d.ut.sum()), flush=True)
print("all modes ran")
