"""Randomised run of Diff_smag2::exec_viscosity on the GPU against the CPU oracle (strain^2, N^2, evisc) on random grid shapes,
surface model on / off, stratified / neutral, with and without the per-level mixing-length table: within the stated 8 ulp.
Test infrastructure (imports tests/ and oracle/)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import common as cm, backends as B
from common import ptr, dbl
from microhh_amd import capi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 9)
be = B.get("hip"); O = cm.oracle()
bad = 0; worst = 0
for it in range(n):
    dtype = np.float64 if rng.random() < 0.6 else np.float32
    sm, neutral = [(1, 0), (0, 0), (1, 1)][int(rng.integers(0, 3))]
    itot = int(rng.choice([4, 7, 16, 33, 64, 65, 70, 96, 130])); jtot = int(rng.choice([3, 4, 5, 9, 12, 17])); ktot = int(rng.integers(6, 40))
    g = cm.grid_2nd(itot, jtot, ktot, gc=(3, 3, 1), dtype=dtype)
    c = cm.Case(g, periodic=True); Gh = g.host_struct()
    cs, tPr, grav = 0.23, 1./3., 9.81
    thref = np.full(g.kcells, 300., dtype=dtype)
    want = np.zeros(g.shape3, dtype=dtype)
    O.orc_smag2_strain2(Gh, sm, ptr(want), ptr(c.u), ptr(c.v), ptr(c.w), ptr(c.dudz), ptr(c.dvdz))
    if neutral:
        O.orc_smag2_evisc_neutral(Gh, sm, ptr(want), ptr(c.u), ptr(c.v), ptr(c.z0m), dbl(cs), dbl(1e-5))
    else:
        n2 = np.zeros(g.shape3, dtype=dtype); O.orc_calc_N2(Gh, ptr(n2), ptr(c.s[0]), ptr(thref), dbl(grav))
        O.orc_smag2_evisc(Gh, sm, ptr(want), ptr(n2), ptr(c.dbdz), ptr(c.z0m), dbl(cs), dbl(tPr))
    for table in (False, True):
        d = B.DevCase(be, c); f = d.fields()
        p = capi.MhhDiffParams(); p.cs = cs; p.tPr = tPr; p.surface_model = sm; p.neutral = neutral
        p.N2 = None; p.th_for_N2 = 0; dthref = be.arr(thref); p.thref = be.ptr(dthref).value; p.grav = grav
        ml_h = np.zeros(g.kcells, dtype=dtype); B.ok(be, be.lib.mhh_smag2_mlen0_host(Gh, cs, ptr(ml_h)))
        ml = be.arr(ml_h); p.mlen0 = be.ptr(ml).value
        if table:
            m2_h = np.zeros(g.kcells, dtype=dtype)
            B.ok(be, be.lib.mhh_smag2_mlen2_host(Gh, sm, neutral, ptr(ml_h), float(c.z0m.flat[0]), ptr(m2_h)))
            m2 = be.arr(m2_h); p.mlen2 = be.ptr(m2).value
        B.ok(be, be.lib.mhh_diff_exec_viscosity(d.G, cm.DIFF_SMAG2, C.byref(f), C.byref(p), be.stream))
        ev = be.host(d.evisc)
        k0, k1 = (g.kstart, g.kend) if sm else (g.kstart-1, g.kend+1)
        u = cm.ulp_diff(ev[k0:k1], want[k0:k1]); worst = max(worst, u)
        if u > 8:
            bad += 1; print("OUT OF TOLERANCE", sm, neutral, g.shape3, np.dtype(dtype).name, table, u, flush=True)
print("evisc fuzz: %d cases x 2 forms, %d beyond 8 ulp, worst %d ulp" % (n, bad, worst))
sys.exit(1 if bad else 0)
