"""Randomised parity run on the GPU: the fused passes and the separate operator calls against the CPU oracle, bit for bit, on
random grid shapes (ragged tiles, short columns, 2-D grids), both precisions. Test infrastructure (imports tests/ and oracle/)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import common as cm, backends as B, test_parity as T
from microhh_amd import capi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
be = B.get("hip")
bad = 0
for it in range(n):
    dtype = np.float64 if rng.random() < 0.6 else np.float32
    pair = rng.integers(0, 2)
    itot = int(rng.choice([4, 7, 16, 33, 64, 65, 70, 96, 128, 130, 140, 256])); jtot = int(rng.choice([1, 3, 4, 5, 9, 12, 17])); ktot = int(rng.integers(6, 40))
    if pair == 0:
        adv, dif, sm = cm.ADVEC_2I5, cm.DIFF_SMAG2, int(rng.integers(0, 2))
        if jtot < 3: jtot = 3
        g = cm.grid_2nd(itot, jtot, ktot, gc=(3, 3, int(rng.integers(1, 3))), dtype=dtype)
    else:
        adv, dif, sm = cm.ADVEC_4, cm.DIFF_4, 0
        if jtot != 1 and jtot < 3: jtot = 3
        g = cm.grid_4th(itot, jtot, ktot, dtype=dtype)
    c = cm.Case(g, nscalars=int(rng.integers(1, 3)), seed=int(rng.integers(0, 1 << 30))) if "seed" in cm.Case.__init__.__code__.co_varnames else cm.Case(g, nscalars=int(rng.integers(1, 3)))
    want = T._oracle_rhs(c, adv, dif, sm)
    p = capi.MhhDiffParams(); p.cs = 0.23; p.tPr = 1./3.; p.surface_model = sm
    for fused in (True, False):
        d = B.DevCase(be, c); f = d.fields()
        if fused:
            B.ok(be, be.lib.mhh_rhs_exec(d.G, adv, dif, C.byref(f), C.byref(p), be.stream))
        else:
            B.ok(be, be.lib.mhh_advec_exec(d.G, adv, C.byref(f), be.stream)); B.ok(be, be.lib.mhh_diff_exec(d.G, dif, C.byref(f), C.byref(p), be.stream))
        got = (be.host(d.ut), be.host(d.vt), be.host(d.wt), [be.host(x) for x in d.st])
        ok = all(np.array_equal(a, b) for a, b in zip(got[:3], want[:3])) and all(np.array_equal(a, b) for a, b in zip(got[3], want[3]))
        if not ok:
            bad += 1; print("MISMATCH", adv, dif, sm, g.shape3, np.dtype(dtype).name, "fused" if fused else "two calls", flush=True)
print("parity fuzz: %d cases x 2 forms, %d mismatches" % (n, bad))
sys.exit(1 if bad else 0)
