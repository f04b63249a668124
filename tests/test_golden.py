"""Golden vectors recorded from the reference's own kernels (tests/golden/make_golden.py, run where /root/reference
exists). The oracle must reproduce them bit for bit everywhere (also on the GPU box, where the reference is
absent); so must the HIP path (gpu) and its CPU emulation."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import backends as B
import common as cm
from common import ptr, dbl

sys.path.insert(0, os.path.join(cm.ROOT, "tests", "golden"))
import make_golden as mg  # noqa: E402

GOLD = np.load(os.path.join(cm.ROOT, "tests", "golden", "ref_vectors.npz"))


def test_input_recipe_unchanged():
    g2, c2, g4, c4 = mg.cases(GOLD["z2"], GOLD["z4"])
    assert np.array_equal(GOLD["fingerprint"], np.array([c2.u.sum(), c2.rhorefh.sum(), c4.w.sum(), c2.evisc.sum()]))


def test_oracle_reproduces_reference_vectors():
    O = cm.oracle()
    g2, c2, g4, c4 = mg.cases(GOLD["z2"], GOLD["z4"])
    for scheme, g, c in ((2, g2, c2), (25, g2, c2), (24, g2, c2), (262, g2, c2), (253, g2, c2), (4, g4, c4), (41, g4, c4)):
        G = g.host_struct()
        for fn, tn in ((O.orc_advec_u, "ut"), (O.orc_advec_v, "vt"), (O.orc_advec_w, "wt")):
            t = c.copy_of(tn); fn(G, scheme, ptr(t), ptr(c.u), ptr(c.v), ptr(c.w), ptr(c.rhoref), ptr(c.rhorefh))
            assert np.array_equal(t[g.interior], GOLD["advec%d_%s" % (scheme, tn)])
        t = c.st[0].copy(); O.orc_advec_s(G, scheme, ptr(t), ptr(c.s[0]), ptr(c.u), ptr(c.v), ptr(c.w), ptr(c.rhoref), ptr(c.rhorefh))
        assert np.array_equal(t[g.interior], GOLD["advec%d_st" % scheme])
        assert O.orc_advec_cfl(G, scheme, ptr(c.u), ptr(c.v), ptr(c.w), dbl(0.37)) == float(GOLD["advec%d_cfl" % scheme])
    for order, g, c in ((2, g2, c2), (4, g4, c4)):
        G = g.host_struct()
        for fn, src, tn in ((O.orc_diff_c, c.u, "ut"), (O.orc_diff_w, c.w, "wt")):
            t = c.copy_of(tn); fn(G, order, ptr(t), ptr(src), dbl(1.3e-2))
            assert np.array_equal(t[g.interior], GOLD["diff%d_%s" % (order, tn)])
    G = g2.host_struct(); c = c2
    for sm in (0, 1):
        s2 = np.zeros(g2.shape3); O.orc_smag2_strain2(G, sm, ptr(s2), ptr(c.u), ptr(c.v), ptr(c.w), ptr(c.dudz), ptr(c.dvdz))
        assert np.array_equal(s2[g2.interior], GOLD["smag%d_strain2" % sm])
        t = c.copy_of("ut"); O.orc_smag2_diff_u(G, sm, ptr(t), ptr(c.u), ptr(c.v), ptr(c.w), ptr(c.evisc), ptr(c.u_fluxbot), ptr(c.u_fluxtop), ptr(c.rhoref), ptr(c.rhorefh), dbl(1e-5))
        assert np.array_equal(t[g2.interior], GOLD["smag%d_ut" % sm])
        t = c.copy_of("vt"); O.orc_smag2_diff_v(G, sm, ptr(t), ptr(c.u), ptr(c.v), ptr(c.w), ptr(c.evisc), ptr(c.v_fluxbot), ptr(c.v_fluxtop), ptr(c.rhoref), ptr(c.rhorefh), dbl(1e-5))
        assert np.array_equal(t[g2.interior], GOLD["smag%d_vt" % sm])
        t = c.copy_of("wt"); O.orc_smag2_diff_w(G, ptr(t), ptr(c.u), ptr(c.v), ptr(c.w), ptr(c.evisc), ptr(c.rhoref), ptr(c.rhorefh), dbl(1e-5))
        assert np.array_equal(t[g2.interior], GOLD["smag%d_wt" % sm])
        t = c.st[0].copy(); O.orc_smag2_diff_c(G, sm, ptr(t), ptr(c.s[0]), ptr(c.evisc), ptr(c.s_fluxbot), ptr(c.s_fluxtop), ptr(c.rhoref), ptr(c.rhorefh), dbl(1./3.), dbl(1e-5))
        assert np.array_equal(t[g2.interior], GOLD["smag%d_st" % sm])
    assert O.orc_smag2_dnmul(G, ptr(c.evisc), dbl(1./3.)) == float(GOLD["smag_dnmul"])
    ul, vl, wl, sl = cm.limiter_inputs(c, np.float64)
    t = c.st[0].copy(); O.orc_advec_s_lim(G, ptr(t), ptr(sl), ptr(ul), ptr(vl), ptr(wl), ptr(c.rhoref), ptr(c.rhorefh))
    assert np.array_equal(t[g2.interior], GOLD["advec_s_lim_st"])


@pytest.mark.parametrize("name", [pytest.param("emul"), pytest.param("hip", marks=pytest.mark.gpu)])
def test_hip_path_reproduces_reference_vectors(name):
    be = B.get(name)
    g2, c2, g4, c4 = mg.cases(GOLD["z2"], GOLD["z4"])
    for scheme, g, c in ((2, g2, c2), (25, g2, c2), (24, g2, c2), (262, g2, c2), (253, g2, c2), (4, g4, c4), (41, g4, c4)):
        d = B.DevCase(be, c)
        for fn, tn in ((be.lib.mhh_advec_u, "ut"), (be.lib.mhh_advec_v, "vt"), (be.lib.mhh_advec_w, "wt")):
            t = be.arr(getattr(c, tn))
            B.ok(be, fn(d.G, scheme, be.ptr(t), be.ptr(d.u), be.ptr(d.v), be.ptr(d.w), be.ptr(d.rhoref), be.ptr(d.rhorefh), be.stream))
            assert np.array_equal(be.host(t)[g.interior], GOLD["advec%d_%s" % (scheme, tn)]), (scheme, tn)
        t = be.arr(c.st[0])
        B.ok(be, be.lib.mhh_advec_s(d.G, scheme, be.ptr(t), be.ptr(d.s[0]), be.ptr(d.u), be.ptr(d.v), be.ptr(d.w), be.ptr(d.rhoref), be.ptr(d.rhorefh), be.stream))
        assert np.array_equal(be.host(t)[g.interior], GOLD["advec%d_st" % scheme])
        out = C.c_double(0)
        B.ok(be, be.lib.mhh_advec_cfl(d.G, scheme, be.ptr(d.u), be.ptr(d.v), be.ptr(d.w), 0.37, be.ptr(d.work), C.byref(out), be.stream))
        assert out.value == float(GOLD["advec%d_cfl" % scheme])
    for order, g, c in ((2, g2, c2), (4, g4, c4)):
        d = B.DevCase(be, c)
        for fn, src, tn in ((be.lib.mhh_diff_c, d.u, "ut"), (be.lib.mhh_diff_w, d.w, "wt")):
            t = be.arr(getattr(c, tn))
            B.ok(be, fn(d.G, order, be.ptr(t), be.ptr(src), 1.3e-2, be.stream))
            assert np.array_equal(be.host(t)[g.interior], GOLD["diff%d_%s" % (order, tn)])
    c = c2; d = B.DevCase(be, c)
    for sm in (0, 1):
        s2 = be.zeros(g2.shape3, np.float64)
        B.ok(be, be.lib.mhh_smag2_strain2(d.G, sm, be.ptr(s2), be.ptr(d.u), be.ptr(d.v), be.ptr(d.w), be.ptr(d.dudz), be.ptr(d.dvdz), be.stream))
        assert np.array_equal(be.host(s2)[g2.interior], GOLD["smag%d_strain2" % sm])
        t = be.arr(c.ut)
        B.ok(be, be.lib.mhh_smag2_diff_u(d.G, sm, be.ptr(t), be.ptr(d.u), be.ptr(d.v), be.ptr(d.w), be.ptr(d.evisc), be.ptr(d.u_fluxbot), be.ptr(d.u_fluxtop), be.ptr(d.rhoref), be.ptr(d.rhorefh), 1e-5, be.stream))
        assert np.array_equal(be.host(t)[g2.interior], GOLD["smag%d_ut" % sm])
        t = be.arr(c.st[0])
        B.ok(be, be.lib.mhh_smag2_diff_c(d.G, sm, be.ptr(t), be.ptr(d.s[0]), be.ptr(d.evisc), be.ptr(d.s_fluxbot), be.ptr(d.s_fluxtop), be.ptr(d.rhoref), be.ptr(d.rhorefh), 1./3., 1e-5, be.stream))
        assert np.array_equal(be.host(t)[g2.interior], GOLD["smag%d_st" % sm])
    ul, vl, wl, sl = cm.limiter_inputs(c, np.float64)
    t = be.arr(c.st[0])
    dl = [be.arr(x) for x in (sl, ul, vl, wl)]          # keep the device copies alive across the call
    B.ok(be, be.lib.mhh_advec_s_lim(d.G, be.ptr(t), *[be.ptr(x) for x in dl], be.ptr(d.rhoref), be.ptr(d.rhorefh), be.stream))
    assert np.array_equal(be.host(t)[g2.interior], GOLD["advec_s_lim_st"])
