"""Pin the oracle: bit-for-bit equality with the reference's own kernels (oracle/_ref, built from the
reference sources in place with identical flags -O2 -ffp-contract=off) on seeded random fields, for
every stencil kernel of SURVEY.md §8a that the reference TU exposes without Grid/FFTW.

Skipped only where neither /root/reference nor a prebuilt oracle/_ref/libmhhref.so exists."""
import numpy as np
import pytest

import common as cm
from common import ptr, dbl

REF = cm.ref()
pytestmark = pytest.mark.skipif(REF is None, reason="oracle/_ref not available (no reference, no prebuilt lib)")

DTYPES = [np.float64, np.float32]


def grids2(dtype):
    return [cm.grid_2nd(*gs[:3], gc=gs[3:], dtype=dtype) for gs in cm.SMALL_GRIDS_2] + \
           [cm.grid_2nd(12, 1, 8, gc=(3, 3, 1), dtype=dtype)]          # 2-D run (jtot == 1)


def grids4(dtype):
    return [cm.grid_4th(16, 12, 12, dtype=dtype), cm.grid_4th(10, 8, 8, dtype=dtype), cm.grid_4th(12, 1, 8, dtype=dtype)]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("scheme,refname", [(cm.ADVEC_2, "ref_advec_2"), (cm.ADVEC_2I5, "ref_advec_2i5"), (cm.ADVEC_2I4, "ref_advec_2i4"), (cm.ADVEC_2I62, "ref_advec_2i62"), (cm.ADVEC_2I53, "ref_advec_2i53"), (cm.ADVEC_4, "ref_advec_4"), (cm.ADVEC_4M, "ref_advec_4m")])
def test_advec_bitwise(scheme, refname, dtype):
    O = cm.oracle()
    for g in (grids4(dtype) if scheme in (cm.ADVEC_4, cm.ADVEC_4M) else grids2(dtype)):
        if scheme == cm.ADVEC_2I5 and g.ktot < 6:
            continue
        if scheme in (cm.ADVEC_2I4, cm.ADVEC_2I53) and (g.igc < 2 or g.jgc < 2 or g.ktot < 4):
            continue
        c = cm.Case(g)
        G = g.host_struct()
        for comp, (oname, tname) in enumerate([("orc_advec_u", "ut"), ("orc_advec_v", "vt"), ("orc_advec_w", "wt")]):
            t_o, t_r = c.copy_of(tname), c.copy_of(tname)
            getattr(O, oname)(G, scheme, ptr(t_o), ptr(c.u), ptr(c.v), ptr(c.w), ptr(c.rhoref), ptr(c.rhorefh))
            getattr(REF, refname)(G, comp, ptr(t_r), None, ptr(c.u), ptr(c.v), ptr(c.w), ptr(c.rhoref), ptr(c.rhorefh))
            assert np.array_equal(t_o, t_r), (refname, comp, g.shape3, cm.ulp_diff(t_o, t_r))
            assert not np.array_equal(t_o, getattr(c, tname))
        t_o, t_r = c.st[0].copy(), c.st[0].copy()
        O.orc_advec_s(G, scheme, ptr(t_o), ptr(c.s[0]), ptr(c.u), ptr(c.v), ptr(c.w), ptr(c.rhoref), ptr(c.rhorefh))
        getattr(REF, refname)(G, 3, ptr(t_r), ptr(c.s[0]), ptr(c.u), ptr(c.v), ptr(c.w), ptr(c.rhoref), ptr(c.rhorefh))
        assert np.array_equal(t_o, t_r), (refname, "s", g.shape3, cm.ulp_diff(t_o, t_r))
        cfl_o = O.orc_advec_cfl(G, scheme, ptr(c.u), ptr(c.v), ptr(c.w), dbl(0.37))
        cfl_r = getattr(REF, refname + "_cfl")(G, ptr(c.u), ptr(c.v), ptr(c.w), dbl(0.37))
        assert cfl_o == cfl_r and cfl_o > 0


@pytest.mark.parametrize("dtype", DTYPES)
def test_advec_s_lim_bitwise(dtype):
    """Koren flux limiter (include/advec_monotonic.h:10-180) against the reference's own template."""
    O = cm.oracle()
    for g in grids2(dtype) + [cm.grid_2nd(12, 8, 8, gc=(3, 3, 2), dtype=dtype)]:
        if g.ktot < 6:
            continue
        c = cm.Case(g)
        G = g.host_struct()
        u, v, w, s = cm.limiter_inputs(c, dtype)
        t_o, t_r = c.st[0].copy(), c.st[0].copy()
        O.orc_advec_s_lim(G, ptr(t_o), ptr(s), ptr(u), ptr(v), ptr(w), ptr(c.rhoref), ptr(c.rhorefh))
        REF.ref_advec_s_lim(G, ptr(t_r), ptr(s), ptr(u), ptr(v), ptr(w), ptr(c.rhoref), ptr(c.rhorefh))
        assert np.array_equal(t_o, t_r), ("advec_s_lim", g.shape3, cm.ulp_diff(t_o, t_r))
        assert not np.array_equal(t_o, c.st[0])


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("order", [2, 4])
def test_diff_bitwise(order, dtype):
    O = cm.oracle()
    for g in (grids4(dtype) if order == 4 else grids2(dtype)):
        c = cm.Case(g)
        G = g.host_struct()
        for is_w, fld, tname in [(0, c.u, "ut"), (1, c.w, "wt"), (0, c.s[0], "vt")]:
            t_o, t_r = c.copy_of(tname), c.copy_of(tname)
            (O.orc_diff_w if is_w else O.orc_diff_c)(G, order, ptr(t_o), ptr(fld), dbl(1.3e-2))
            getattr(REF, "ref_diff_%d" % order)(G, is_w, ptr(t_r), ptr(fld), dbl(1.3e-2))
            assert np.array_equal(t_o, t_r), (order, is_w, g.shape3, cm.ulp_diff(t_o, t_r))
            assert not np.array_equal(t_o, getattr(c, tname))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("sm", [0, 1])
def test_smag2_bitwise(sm, dtype):
    O = cm.oracle()
    for g in grids2(dtype):
        if g.jtot == 1:
            continue
        c = cm.Case(g)
        G = g.host_struct()
        s_o = np.zeros(g.shape3, dtype=dtype); s_r = np.zeros(g.shape3, dtype=dtype)
        O.orc_smag2_strain2(G, sm, ptr(s_o), ptr(c.u), ptr(c.v), ptr(c.w), ptr(c.dudz), ptr(c.dvdz))
        REF.ref_smag2_strain2(G, sm, ptr(s_r), ptr(c.u), ptr(c.v), ptr(c.w), ptr(c.dudz), ptr(c.dvdz))
        assert np.array_equal(s_o, s_r) and s_o[g.interior].min() > 0
        for comp, (oname, tname, fb, ft) in enumerate([("orc_smag2_diff_u", "ut", c.u_fluxbot, c.u_fluxtop),
                                                       ("orc_smag2_diff_v", "vt", c.v_fluxbot, c.v_fluxtop)]):
            t_o, t_r = c.copy_of(tname), c.copy_of(tname)
            getattr(O, oname)(G, sm, ptr(t_o), ptr(c.u), ptr(c.v), ptr(c.w), ptr(c.evisc), ptr(fb), ptr(ft), ptr(c.rhoref), ptr(c.rhorefh), dbl(1e-5))
            REF.ref_smag2_diff_uvw(G, comp, sm, ptr(t_r), ptr(c.u), ptr(c.v), ptr(c.w), ptr(c.evisc), ptr(fb), ptr(ft), ptr(c.rhoref), ptr(c.rhorefh), dbl(1e-5))
            assert np.array_equal(t_o, t_r), (oname, sm, cm.ulp_diff(t_o, t_r))
        t_o, t_r = c.copy_of("wt"), c.copy_of("wt")
        O.orc_smag2_diff_w(G, ptr(t_o), ptr(c.u), ptr(c.v), ptr(c.w), ptr(c.evisc), ptr(c.rhoref), ptr(c.rhorefh), dbl(1e-5))
        REF.ref_smag2_diff_uvw(G, 2, sm, ptr(t_r), ptr(c.u), ptr(c.v), ptr(c.w), ptr(c.evisc), None, None, ptr(c.rhoref), ptr(c.rhorefh), dbl(1e-5))
        assert np.array_equal(t_o, t_r)
        t_o, t_r = c.st[0].copy(), c.st[0].copy()
        O.orc_smag2_diff_c(G, sm, ptr(t_o), ptr(c.s[0]), ptr(c.evisc), ptr(c.s_fluxbot), ptr(c.s_fluxtop), ptr(c.rhoref), ptr(c.rhorefh), dbl(1./3.), dbl(1e-5))
        REF.ref_smag2_diff_c(G, sm, ptr(t_r), ptr(c.s[0]), ptr(c.evisc), ptr(c.s_fluxbot), ptr(c.s_fluxtop), ptr(c.rhoref), ptr(c.rhorefh), dbl(1./3.), dbl(1e-5))
        assert np.array_equal(t_o, t_r)
        for tPr in (1./3., 1.7):
            assert O.orc_smag2_dnmul(G, ptr(c.evisc), dbl(tPr)) == REF.ref_smag2_dnmul(G, ptr(c.evisc), dbl(tPr))
