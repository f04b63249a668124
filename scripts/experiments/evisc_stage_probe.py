import os, sys, ctypes as C
ROOT="/root/repo"; sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import common as cm, backends as B
from common import ptr, dbl
from microhh_amd import capi
dtype=np.float64; sm, neutral = 1, 0
g = cm.grid_2nd(40, 11, 39, gc=(3, 3, 1), dtype=dtype)
c = cm.Case(g, periodic=True); Gh = g.host_struct()
cs, tPr, grav = 0.23, 1./3., 9.81
thref = np.full(g.kcells, 300., dtype=dtype)
res = {}
for name in ("hip", "emul"):
    be = B.get(name)
    d = B.DevCase(be, c); f = d.fields()
    s2 = be.zeros(g.shape3, dtype)
    B.ok(be, be.lib.mhh_smag2_strain2(d.G, sm, be.ptr(s2), be.ptr(d.u), be.ptr(d.v), be.ptr(d.w), be.ptr(d.dudz), be.ptr(d.dvdz), be.stream))
    n2 = be.zeros(g.shape3, dtype); dthref = be.arr(thref)
    B.ok(be, be.lib.mhh_calc_N2(d.G, be.ptr(n2), be.ptr(d.s[0]), be.ptr(dthref), grav, be.stream))
    ml = B.mlen0(be, g, cs)
    ev = be.arr(be.host(s2))
    B.ok(be, be.lib.mhh_smag2_evisc(d.G, sm, be.ptr(ev), be.ptr(n2), be.ptr(d.dbdz), be.ptr(d.z0m), be.ptr(ml), tPr, be.stream))
    p = capi.MhhDiffParams(); p.cs = cs; p.tPr = tPr; p.surface_model = sm; p.neutral = neutral
    p.N2 = None; p.th_for_N2 = 0; p.thref = be.ptr(dthref).value; p.grav = grav; p.mlen0 = be.ptr(ml).value
    B.ok(be, be.lib.mhh_diff_exec_viscosity(d.G, cm.DIFF_SMAG2, C.byref(f), C.byref(p), be.stream))
    res[name] = dict(s2=be.host(s2), n2=be.host(n2), ev3=be.host(ev), evf=be.host(d.evisc))
sl=(slice(g.kstart,g.kend),slice(g.jstart,g.jend),slice(g.istart,g.iend))
for k in ("s2","n2","ev3","evf"):
    a=res["hip"][k][sl]; b=res["emul"][k][sl]
    u=np.abs(a.view(np.int64)-b.view(np.int64))
    print(k, "hip vs emul: max ulp", u.max(), "cells differing", (u>0).sum(), "at k levels", np.unique(np.nonzero(u)[0])[:10])
