"""Randomised run of Pres::exec on the GPU against the CPU oracle (input -> spectral solve -> output) on random grid shapes,
pres_2 and pres_4, both precisions: p and the corrected tendencies within the stated tolerance (1e-11 / 2e-4 of max|p|).
A third argument "lds" draws power-of-two grids (pres_2 and pres_4) and forces the form with the transforms in LDS (csrc/pres_lds.h) with a random
number of levels per block; p is then compared with all its ghost cells. Test infrastructure (imports tests/ and oracle/)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import common as cm, backends as B
from common import ptr, dbl
from microhh_amd import capi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
lds = len(sys.argv) > 3 and sys.argv[3] == "lds"
be = B.get("hip"); O = cm.oracle()
bad = 0; worst = 0.0
for it in range(n):
    dtype = np.float64 if rng.random() < 0.6 else np.float32
    order = 2 if rng.random() < 0.6 else 4
    itot = int(rng.choice([4, 6, 8, 12, 16, 20, 30, 32, 48, 64])); jtot = int(rng.choice([1, 3, 4, 6, 8, 10, 16, 24])); ktot = int(rng.integers(4, 24))
    if lds:
        # every instantiated row length (csrc/k_pres.hip): itot 16 ... 1024, jtot 8 ... 1024 (fp64: ... 512); few levels on the big planes
        itot = int(rng.choice([16, 32, 64, 128, 256, 512, 1024])); jtot = int(rng.choice([8, 16, 32, 64, 128, 256, 512, 1024]))
        if (dtype == np.float64 or order == 4) and jtot == 1024: jtot = 512         # no instantiation (it would need scratch): staged form
        ktot = int(rng.integers(2, 40)) if itot*jtot <= 128*128 else int(rng.integers(2, 9))
        os.environ["MHH_PRES_LDS"] = "1"; os.environ["MHH_PRES_LDS_KC"] = str(int(rng.integers(1, 12)))
    if order == 2:
        g = cm.grid_2nd(itot, jtot, ktot, gc=(int(rng.integers(1, 4)), int(rng.integers(1, 4)), 1), dtype=dtype)
        c = cm.Case(g, rho="random", periodic=True)
    else:
        if jtot != 1 and jtot < 4: jtot = 4
        if itot < 6: itot = 6
        g = cm.grid_4th(itot, jtot, max(ktot, 6), dtype=dtype)
        c = cm.Case(g, rho="one", periodic=True)
        for m in (1, 2):
            c.w[g.kstart-m] = -c.w[g.kstart+m]; c.w[g.kend+m] = -c.w[g.kend-m]
    Gh = g.host_struct(); dt = 0.7
    pk = np.zeros((g.ktot, g.jtot, g.itot), dtype=dtype)
    ut, vt, wt = c.ut.copy(), c.vt.copy(), c.wt.copy()
    O.orc_pres_input(Gh, order, ptr(pk), ptr(c.u), ptr(c.v), ptr(c.w), ptr(ut), ptr(vt), ptr(wt), ptr(c.rhoref), ptr(c.rhorefh), dbl(dt))
    p_want = np.zeros(g.shape3, dtype=dtype)
    O.orc_pres_solve(Gh, order, ptr(p_want), ptr(pk), ptr(c.rhoref), ptr(c.rhorefh))
    O.orc_pres_output(Gh, order, ptr(ut), ptr(vt), ptr(wt), ptr(p_want))
    d = B.DevCase(be, c); f = d.fields()
    plan = capi.PLAN()
    B.ok(be, be.lib.mhh_pres_plan_create(Gh, order, ptr(g.dz), ptr(g.dzhi), ptr(g.dzi4), ptr(g.dzhi4), ptr(c.rhoref), ptr(c.rhorefh), C.byref(plan)))
    B.ok(be, be.lib.mhh_pres_exec(plan, d.G, C.byref(f), dt, be.stream))
    sl = (slice(g.kstart, g.kend), slice(g.jstart, g.jend), slice(g.istart, g.iend))
    if lds:
        assert be.lib.mhh_pres_exec_form(plan) == 1
        sl = (slice(g.kstart-1, g.kend), slice(None), slice(None))          # p with its periodic halo and the ghost level below
        if order == 4: sl = (slice(g.kstart-2, g.kend+2), slice(None), slice(None))      # ... its four mirrored ghost levels
    tol = 1e-11 if dtype == np.float64 else 2e-4
    scale = np.abs(p_want[sl]).max()
    err = np.abs(be.host(d.p)[sl] - p_want[sl]).max() / scale
    sl = (slice(g.kstart, g.kend), slice(g.jstart, g.jend), slice(g.istart, g.iend))
    gscale = max(np.abs(ut[sl] - c.ut[sl]).max(), np.abs(vt[sl] - c.vt[sl]).max(), np.abs(wt[sl] - c.wt[sl]).max())
    terr = max(np.abs(be.host(x)[sl] - w_[sl]).max() for x, w_ in ((d.ut, ut), (d.vt, vt), (d.wt, wt))) / gscale
    be.lib.mhh_pres_plan_destroy(plan)
    worst = max(worst, err / tol, terr / (10*tol))
    if err > tol or terr > 10*tol:
        bad += 1; print("OUT OF TOLERANCE", order, g.shape3, np.dtype(dtype).name, err, terr, flush=True)
print("pressure fuzz: %d cases, %d out of tolerance, worst error / tolerance %.3f" % (n, bad, worst))
sys.exit(1 if bad else 0)
