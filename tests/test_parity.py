"""Parity of the HIP path (through the C ABI) with the CPU oracle on identical seeded inputs.

Every test runs on two backends (tests/backends.py): ``hip`` = the product on a real MI355X (marked gpu) and
``emul`` = the same kernel sources executed on the CPU by the test-only HIP stand-in.

Tolerances (fp64 and fp32 alike, stated per test):
  * stencil operators (advec_*, diff_*, smag2 strain/diff, pres in/out, cfl/dnmul/div maxima, rk, cyclic):
    BIT-EXACT -- the library is built with -ffp-contract=off and keeps the reference's expression association;
  * evisc: <= 8 ulp (device sqrt and division are correctly rounded -- scripts/experiments/ieee_probe.hip --, the oracle follows the
    reference's pow(x,2) / pow(y,.5), whose last bits depend on the host's libm: up to 6 ulp apart in 3500 fuzz cases);
  * van-Driest evisc (pow .25, exp): <= 64 ulp;
  * pressure solve: |dp| <= 1e-11 max|p| fp64 / 2e-4 fp32 (rocFFT vs the oracle's DFT; complex vs half-complex solve).
"""
import ctypes as C
import os

import numpy as np
import pytest

import backends as B
import common as cm
from common import ptr, dbl
from microhh_amd import capi

BACKENDS = [pytest.param("emul"), pytest.param("hip", marks=pytest.mark.gpu)]
DTYPES = [np.float64, np.float32]


@pytest.fixture(params=BACKENDS)
def be(request):
    return B.get(request.param)


def grids2(dtype, small=False):
    gs = [cm.grid_2nd(16, 12, 10, gc=(3, 3, 1), dtype=dtype), cm.grid_2nd(70, 9, 8, gc=(3, 3, 2), dtype=dtype),
          cm.grid_2nd(12, 1, 8, gc=(3, 3, 1), dtype=dtype)]
    return gs[:1] if small else gs


def grids4(dtype, small=False):
    gs = [cm.grid_4th(16, 12, 12, dtype=dtype), cm.grid_4th(66, 5, 8, dtype=dtype), cm.grid_4th(12, 1, 8, dtype=dtype)]
    return gs[:1] if small else gs


def same(a, b):
    return np.array_equal(a, b)


# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
def test_boundary_cyclic(be, dtype):
    O = cm.oracle()
    for g in grids2(dtype) + grids4(dtype, small=True):
        c = cm.Case(g)
        for edge in (cm.EDGE_EW, cm.EDGE_NS, cm.EDGE_BOTH):
            want = c.u.copy(); O.orc_boundary_cyclic(g.host_struct(), ptr(want), edge)
            d = be.arr(c.u)
            B.ok(be, be.lib.mhh_boundary_cyclic(be.grid(g), be.ptr(d), edge, be.stream))
            assert same(be.host(d), want), (g.shape3, edge)
        # several fields per launch
        ds = [be.arr(c.u), be.arr(c.v), be.arr(c.w)]
        arr = (C.c_void_p * 3)(*[be.ptr(x).value for x in ds])
        B.ok(be, be.lib.mhh_boundary_cyclic_n(be.grid(g), arr, 3, cm.EDGE_BOTH, be.stream))
        for x, src in zip(ds, (c.u, c.v, c.w)):
            want = src.copy(); O.orc_boundary_cyclic(g.host_struct(), ptr(want), cm.EDGE_BOTH)
            assert same(be.host(x), want)
        # 2-D slice
        want = c.dudz.copy(); O.orc_boundary_cyclic_2d(g.host_struct(), ptr(want))
        d = be.arr(c.dudz)
        B.ok(be, be.lib.mhh_boundary_cyclic_2d(be.grid(g), be.ptr(d), be.stream))
        assert same(be.host(d), want)


def test_boundary_cyclic_unsigned_int(be):
    """Boundary_cyclic::exec(unsigned int*) / exec_2d(unsigned int*) (src/boundary_cyclic.cxx:510-660): the integer masks are filled
    like a field of the same shape -- checked against the oracle's fill of the same bit patterns viewed as float32 (a copy)."""
    O = cm.oracle()
    for gd in (np.float64, np.float32):
        g = cm.grid_2nd(16, 12, 10, gc=(3, 3, 1), dtype=gd)
        g32 = cm.grid_2nd(16, 12, 10, gc=(3, 3, 1), dtype=np.float32)
        rs = np.random.RandomState(5)
        a = rs.randint(0, 2**32, size=g.shape3, dtype=np.uint64).astype(np.uint32)
        a2 = rs.randint(0, 2**32, size=g.shape2, dtype=np.uint64).astype(np.uint32)
        for edge in (cm.EDGE_EW, cm.EDGE_NS, cm.EDGE_BOTH):
            want = a.copy().view(np.float32); O.orc_boundary_cyclic(g32.host_struct(), ptr(want), edge)
            d = be.arr(a.view(np.int32))
            B.ok(be, be.lib.mhh_boundary_cyclic_u32(be.grid(g), be.ptr(d), edge, be.stream))
            assert np.array_equal(be.host(d).view(np.uint32), want.view(np.uint32)) and not np.array_equal(want.view(np.uint32), a)
        want = a2.copy().view(np.float32); O.orc_boundary_cyclic_2d(g32.host_struct(), ptr(want))
        d = be.arr(a2.view(np.int32))
        B.ok(be, be.lib.mhh_boundary_cyclic_2d_u32(be.grid(g), be.ptr(d), be.stream))
        assert np.array_equal(be.host(d).view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("dtype", DTYPES)
def test_advec_s_lim_bitexact(be, dtype):
    """Flux-limited scalar advection (include/advec_monotonic.h:79-180): kernel and its place in Advec::exec and the fused RHS."""
    O = cm.oracle()
    for g in grids2(dtype) + [cm.grid_2nd(12, 8, 8, gc=(3, 3, 2), dtype=dtype)]:
        if g.ktot < 6:
            continue
        c = cm.Case(g); Gh = g.host_struct()
        c.u, c.v, c.w, c.s[0] = cm.limiter_inputs(c, dtype)
        d = B.DevCase(be, c)
        want = c.st[0].copy()
        O.orc_advec_s_lim(Gh, ptr(want), ptr(c.s[0]), ptr(c.u), ptr(c.v), ptr(c.w), ptr(c.rhoref), ptr(c.rhorefh))
        t = be.arr(c.st[0])
        B.ok(be, be.lib.mhh_advec_s_lim(d.G, be.ptr(t), be.ptr(d.s[0]), be.ptr(d.u), be.ptr(d.v), be.ptr(d.w), be.ptr(d.rhoref), be.ptr(d.rhorefh), be.stream))
        got = be.host(t)
        assert same(got, want), ("advec_s_lim", g.shape3, cm.ulp_diff(got, want))
        # Advec_2i5::exec with the scalar in fluxlimit_list (src/advec_2i5.cxx:921,1030)
        f = d.fields(); f.s_fluxlimit[0] = 1
        B.ok(be, be.lib.mhh_advec_exec(d.G, cm.ADVEC_2I5, C.byref(f), be.stream))
        assert same(be.host(d.st[0]), want), ("advec_exec with limiter", g.shape3)
        wu = c.copy_of("ut"); O.orc_advec_u(Gh, cm.ADVEC_2I5, ptr(wu), ptr(c.u), ptr(c.v), ptr(c.w), ptr(c.rhoref), ptr(c.rhorefh))
        assert same(be.host(d.ut), wu)
        # the limiter belongs to advec_2i5 only
        assert be.lib.mhh_advec_exec(d.G, cm.ADVEC_2, C.byref(f), be.stream) != 0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("scheme", [cm.ADVEC_2, cm.ADVEC_2I5, cm.ADVEC_2I4, cm.ADVEC_2I62, cm.ADVEC_2I53, cm.ADVEC_4, cm.ADVEC_4M])
def test_advec_kernels_bitexact(be, scheme, dtype):
    O = cm.oracle()
    for g in (grids4(dtype) if scheme in (cm.ADVEC_4, cm.ADVEC_4M) else grids2(dtype)):
        if scheme in (cm.ADVEC_2I4, cm.ADVEC_2I53) and (g.igc < 2 or g.jgc < 2 or g.ktot < 4):
            continue
        c = cm.Case(g); d = B.DevCase(be, c); Gh = g.host_struct()
        for oname, hname, tname in [("orc_advec_u", "mhh_advec_u", "ut"), ("orc_advec_v", "mhh_advec_v", "vt"), ("orc_advec_w", "mhh_advec_w", "wt")]:
            want = c.copy_of(tname)
            getattr(O, oname)(Gh, scheme, ptr(want), ptr(c.u), ptr(c.v), ptr(c.w), ptr(c.rhoref), ptr(c.rhorefh))
            t = be.arr(getattr(c, tname))
            B.ok(be, getattr(be.lib, hname)(d.G, scheme, be.ptr(t), be.ptr(d.u), be.ptr(d.v), be.ptr(d.w), be.ptr(d.rhoref), be.ptr(d.rhorefh), be.stream))
            got = be.host(t)
            assert same(got, want), (hname, scheme, g.shape3, cm.ulp_diff(got, want))
        want = c.st[0].copy()
        O.orc_advec_s(Gh, scheme, ptr(want), ptr(c.s[0]), ptr(c.u), ptr(c.v), ptr(c.w), ptr(c.rhoref), ptr(c.rhorefh))
        t = be.arr(c.st[0])
        B.ok(be, be.lib.mhh_advec_s(d.G, scheme, be.ptr(t), be.ptr(d.s[0]), be.ptr(d.u), be.ptr(d.v), be.ptr(d.w), be.ptr(d.rhoref), be.ptr(d.rhorefh), be.stream))
        assert same(be.host(t), want), ("advec_s", scheme, g.shape3)
        out = C.c_double(0)
        B.ok(be, be.lib.mhh_advec_cfl(d.G, scheme, be.ptr(d.u), be.ptr(d.v), be.ptr(d.w), 0.37, be.ptr(d.work), C.byref(out), be.stream))
        assert out.value == O.orc_advec_cfl(Gh, scheme, ptr(c.u), ptr(c.v), ptr(c.w), dbl(0.37)) and out.value > 0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("order", [2, 4])
def test_diff_kernels_bitexact(be, order, dtype):
    O = cm.oracle()
    for g in (grids4(dtype) if order == 4 else grids2(dtype)):
        c = cm.Case(g); d = B.DevCase(be, c); Gh = g.host_struct()
        for is_w, src, dsrc, tname in [(0, c.u, d.u, "ut"), (1, c.w, d.w, "wt"), (0, c.s[0], d.s[0], "vt")]:
            want = c.copy_of(tname)
            (O.orc_diff_w if is_w else O.orc_diff_c)(Gh, order, ptr(want), ptr(src), dbl(1.3e-2))
            t = be.arr(getattr(c, tname))
            B.ok(be, (be.lib.mhh_diff_w if is_w else be.lib.mhh_diff_c)(d.G, order, be.ptr(t), be.ptr(dsrc), 1.3e-2, be.stream))
            got = be.host(t)
            assert same(got, want), (order, is_w, g.shape3, cm.ulp_diff(got, want))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("sm", [0, 1])
def test_smag2_kernels(be, sm, dtype):
    O = cm.oracle()
    for g in grids2(dtype)[:2]:
        c = cm.Case(g); d = B.DevCase(be, c); Gh = g.host_struct()
        cs, tPr, visc = 0.23, 1./3., 1e-5
        # strain^2: bit-exact
        want = np.zeros(g.shape3, dtype=dtype)
        O.orc_smag2_strain2(Gh, sm, ptr(want), ptr(c.u), ptr(c.v), ptr(c.w), ptr(c.dudz), ptr(c.dvdz))
        s2 = be.zeros(g.shape3, dtype)
        B.ok(be, be.lib.mhh_smag2_strain2(d.G, sm, be.ptr(s2), be.ptr(d.u), be.ptr(d.v), be.ptr(d.w), be.ptr(d.dudz), be.ptr(d.dvdz), be.stream))
        assert same(be.host(s2), want)
        # evisc from that strain^2: <= 8 ulp, and periodic halo + wall mirror set
        ml = B.mlen0(be, g, cs)
        ev_want = want.copy()
        O.orc_smag2_evisc(Gh, sm, ptr(ev_want), ptr(c.N2), ptr(c.dbdz), ptr(c.z0m), dbl(cs), dbl(tPr))
        B.ok(be, be.lib.mhh_smag2_evisc(d.G, sm, be.ptr(s2), be.ptr(d.N2), be.ptr(d.dbdz), be.ptr(d.z0m), be.ptr(ml), tPr, be.stream))
        ev = be.host(s2)
        k0, k1 = (g.kstart, g.kend) if sm else (g.kstart-1, g.kend+1)
        assert cm.ulp_diff(ev[k0:k1], ev_want[k0:k1]) <= 8, cm.ulp_diff(ev[k0:k1], ev_want[k0:k1])
        # neutral variants
        evn_want = want.copy()
        O.orc_smag2_evisc_neutral(Gh, sm, ptr(evn_want), ptr(c.u), ptr(c.v), ptr(c.z0m), dbl(cs), dbl(visc))
        s3 = be.arr(want)
        B.ok(be, be.lib.mhh_smag2_evisc_neutral(d.G, sm, be.ptr(s3), be.ptr(d.u), be.ptr(d.v), be.ptr(d.z0m), be.ptr(ml), visc, be.stream))
        evn = be.host(s3)
        assert cm.ulp_diff(evn[k0:k1], evn_want[k0:k1]) <= (4 if sm else 64)
        # stress divergence with a given evisc: bit-exact
        for oname, hname, tname, fb, ft, dfb, dft in [("orc_smag2_diff_u", "mhh_smag2_diff_u", "ut", c.u_fluxbot, c.u_fluxtop, d.u_fluxbot, d.u_fluxtop),
                                                      ("orc_smag2_diff_v", "mhh_smag2_diff_v", "vt", c.v_fluxbot, c.v_fluxtop, d.v_fluxbot, d.v_fluxtop)]:
            w_ = c.copy_of(tname)
            getattr(O, oname)(Gh, sm, ptr(w_), ptr(c.u), ptr(c.v), ptr(c.w), ptr(c.evisc), ptr(fb), ptr(ft), ptr(c.rhoref), ptr(c.rhorefh), dbl(visc))
            t = be.arr(getattr(c, tname))
            B.ok(be, getattr(be.lib, hname)(d.G, sm, be.ptr(t), be.ptr(d.u), be.ptr(d.v), be.ptr(d.w), be.ptr(d.evisc), be.ptr(dfb), be.ptr(dft), be.ptr(d.rhoref), be.ptr(d.rhorefh), visc, be.stream))
            assert same(be.host(t), w_), hname
        w_ = c.copy_of("wt")
        O.orc_smag2_diff_w(Gh, ptr(w_), ptr(c.u), ptr(c.v), ptr(c.w), ptr(c.evisc), ptr(c.rhoref), ptr(c.rhorefh), dbl(visc))
        t = be.arr(c.wt)
        B.ok(be, be.lib.mhh_smag2_diff_w(d.G, be.ptr(t), be.ptr(d.u), be.ptr(d.v), be.ptr(d.w), be.ptr(d.evisc), be.ptr(d.rhoref), be.ptr(d.rhorefh), visc, be.stream))
        assert same(be.host(t), w_)
        w_ = c.st[0].copy()
        O.orc_smag2_diff_c(Gh, sm, ptr(w_), ptr(c.s[0]), ptr(c.evisc), ptr(c.s_fluxbot), ptr(c.s_fluxtop), ptr(c.rhoref), ptr(c.rhorefh), dbl(tPr), dbl(visc))
        t = be.arr(c.st[0])
        B.ok(be, be.lib.mhh_smag2_diff_c(d.G, sm, be.ptr(t), be.ptr(d.s[0]), be.ptr(d.evisc), be.ptr(d.s_fluxbot), be.ptr(d.s_fluxtop), be.ptr(d.rhoref), be.ptr(d.rhorefh), tPr, visc, be.stream))
        assert same(be.host(t), w_)
        out = C.c_double(0)
        for tp in (1./3., 1.7):
            B.ok(be, be.lib.mhh_smag2_dnmul(d.G, be.ptr(d.evisc), tp, be.ptr(d.work), C.byref(out), be.stream))
            assert out.value == O.orc_smag2_dnmul(Gh, ptr(c.evisc), dbl(tp))
        # N2 hook
        thref = (300. + np.arange(g.kcells)).astype(dtype)
        w_ = np.zeros(g.shape3, dtype=dtype); O.orc_calc_N2(Gh, ptr(w_), ptr(c.s[0]), ptr(thref), dbl(9.81))
        n2 = be.zeros(g.shape3, dtype)
        B.ok(be, be.lib.mhh_calc_N2(d.G, be.ptr(n2), be.ptr(d.s[0]), be.ptr(be.arr(thref)), 9.81, be.stream))
        assert same(be.host(n2), w_)


def _oracle_rhs(c, adv, dif, sm, tPr=1./3., visc=1e-5, svisc=1e-5, limited=(), buoy=None):
    """Advec::exec followed by Diff::exec on the oracle; returns the tendencies."""
    O = cm.oracle(); g = c.grid; Gh = g.host_struct()
    ut, vt, wt, st = c.ut.copy(), c.vt.copy(), c.wt.copy(), [x.copy() for x in c.st]
    a = (ptr(c.u), ptr(c.v), ptr(c.w), ptr(c.rhoref), ptr(c.rhorefh))
    if buoy is not None:          # Thermo_dry::exec runs before Advec::exec (src/model.cxx:365,388)
        order, n, threfh, grav = buoy
        O.orc_buoyancy_tend(Gh, order, ptr(wt), ptr(c.s[n]), ptr(threfh), dbl(grav))
    O.orc_advec_u(Gh, adv, ptr(ut), *a); O.orc_advec_v(Gh, adv, ptr(vt), *a); O.orc_advec_w(Gh, adv, ptr(wt), *a)
    for n in range(len(st)):
        if n in limited:
            O.orc_advec_s_lim(Gh, ptr(st[n]), ptr(c.s[n]), *a)
        else:
            O.orc_advec_s(Gh, adv, ptr(st[n]), ptr(c.s[n]), *a)
    if dif in (cm.DIFF_2, cm.DIFF_4):
        o = 2 if dif == cm.DIFF_2 else 4
        O.orc_diff_c(Gh, o, ptr(ut), ptr(c.u), dbl(visc)); O.orc_diff_c(Gh, o, ptr(vt), ptr(c.v), dbl(visc)); O.orc_diff_w(Gh, o, ptr(wt), ptr(c.w), dbl(visc))
        for n in range(len(st)):
            O.orc_diff_c(Gh, o, ptr(st[n]), ptr(c.s[n]), dbl(svisc))
    else:
        O.orc_smag2_diff_u(Gh, sm, ptr(ut), ptr(c.u), ptr(c.v), ptr(c.w), ptr(c.evisc), ptr(c.u_fluxbot), ptr(c.u_fluxtop), ptr(c.rhoref), ptr(c.rhorefh), dbl(visc))
        O.orc_smag2_diff_v(Gh, sm, ptr(vt), ptr(c.u), ptr(c.v), ptr(c.w), ptr(c.evisc), ptr(c.v_fluxbot), ptr(c.v_fluxtop), ptr(c.rhoref), ptr(c.rhorefh), dbl(visc))
        O.orc_smag2_diff_w(Gh, ptr(wt), ptr(c.u), ptr(c.v), ptr(c.w), ptr(c.evisc), ptr(c.rhoref), ptr(c.rhorefh), dbl(visc))
        for n in range(len(st)):
            O.orc_smag2_diff_c(Gh, sm, ptr(st[n]), ptr(c.s[n]), ptr(c.evisc), ptr(c.s_fluxbot), ptr(c.s_fluxtop), ptr(c.rhoref), ptr(c.rhorefh), dbl(tPr), dbl(svisc))
    return ut, vt, wt, st


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("adv,dif,sm", [(cm.ADVEC_2, cm.DIFF_2, 0), (cm.ADVEC_2I5, cm.DIFF_SMAG2, 0), (cm.ADVEC_2I5, cm.DIFF_SMAG2, 1), (cm.ADVEC_4, cm.DIFF_4, 0)])
def test_operator_exec_and_fused_rhs_bitexact(be, adv, dif, sm, dtype):
    """Advec::exec + Diff::exec (unfused entry points) and the fused mhh_rhs_exec give the oracle's bits."""
    for g in (grids4(dtype) if adv == cm.ADVEC_4 else grids2(dtype)):
        if adv == cm.ADVEC_2I5 and g.jtot == 1 and dif == cm.DIFF_SMAG2 and False:
            continue
        for nsc, rho in ((1, "random"), (2, "random"), (1, "one")):      # rho == 1 takes the kernels' division-free path
            c = cm.Case(g, nscalars=nsc, rho=rho)
            want = _oracle_rhs(c, adv, dif, sm)
            p = capi.MhhDiffParams(); p.cs = 0.23; p.tPr = 1./3.; p.surface_model = sm
            # unfused
            d = B.DevCase(be, c); f = d.fields()
            B.ok(be, be.lib.mhh_advec_exec(d.G, adv, C.byref(f), be.stream))
            B.ok(be, be.lib.mhh_diff_exec(d.G, dif, C.byref(f), C.byref(p), be.stream))
            got = (be.host(d.ut), be.host(d.vt), be.host(d.wt), [be.host(x) for x in d.st])
            for a, b, nm in zip(got[:3], want[:3], "uvw"):
                assert same(a, b), ("unfused", nm, adv, dif, g.shape3, cm.ulp_diff(a, b))
            for a, b in zip(got[3], want[3]):
                assert same(a, b), ("unfused s", adv, dif)
            # fused
            d = B.DevCase(be, c); f = d.fields()
            B.ok(be, be.lib.mhh_rhs_exec(d.G, adv, dif, C.byref(f), C.byref(p), be.stream))
            got = (be.host(d.ut), be.host(d.vt), be.host(d.wt), [be.host(x) for x in d.st])
            for a, b, nm in zip(got[:3], want[:3], "uvw"):
                assert same(a, b), ("fused", nm, adv, dif, g.shape3, cm.ulp_diff(a, b))
            for a, b in zip(got[3], want[3]):
                assert same(a, b), ("fused s", adv, dif)


@pytest.mark.parametrize("dtype", DTYPES)
def test_fused_rhs_and_viscosity_with_sixteen_ghost_cells_in_x(be, dtype):
    """A grid whose operators asked for more ghost cells (Grid::set_minimum_ghost_cells, src/grid.cxx:435-439, as
    src/advec_2i5.cxx:42-45 does): igc = 16 makes rows of 512 + 32 cells whole 128-byte lines with istart on a line. Same
    operators, same bits; the marching kernels take their tile origins from istart."""
    adv, dif, sm = cm.ADVEC_2I5, cm.DIFF_SMAG2, 1
    O = cm.oracle()
    for g in (cm.grid_2nd(70, 9, 10, gc=(16, 3, 1), dtype=dtype), cm.grid_2nd(128, 6, 8, gc=(16, 3, 1), dtype=dtype), cm.grid_2nd(16, 5, 8, gc=(4, 3, 1), dtype=dtype)):
        for rho in ("one", "random"):
            c = cm.Case(g, nscalars=1, rho=rho, periodic=True)
            want = _oracle_rhs(c, adv, dif, sm)
            p = capi.MhhDiffParams(); p.cs = 0.23; p.tPr = 1./3.; p.surface_model = sm
            d = B.DevCase(be, c); f = d.fields()
            B.ok(be, be.lib.mhh_rhs_exec(d.G, adv, dif, C.byref(f), C.byref(p), be.stream))
            got = (be.host(d.ut), be.host(d.vt), be.host(d.wt), [be.host(x) for x in d.st])
            for a, b, nm in zip(got[:3], want[:3], "uvw"):
                assert same(a, b), ("fused", nm, g.shape3, rho, cm.ulp_diff(a, b))
            assert same(got[3][0], want[3][0]), ("fused s", g.shape3, rho)
        # exec_viscosity on the same layout: marching form == cell form is tested elsewhere; here against the oracle
        c = cm.Case(g, periodic=True); Gh = g.host_struct()
        thref = np.full(g.kcells, 300., dtype=dtype)
        want = np.zeros(g.shape3, dtype=dtype); n2 = np.zeros(g.shape3, dtype=dtype)
        O.orc_smag2_strain2(Gh, sm, ptr(want), ptr(c.u), ptr(c.v), ptr(c.w), ptr(c.dudz), ptr(c.dvdz))
        O.orc_calc_N2(Gh, ptr(n2), ptr(c.s[0]), ptr(thref), dbl(9.81))
        O.orc_smag2_evisc(Gh, sm, ptr(want), ptr(n2), ptr(c.dbdz), ptr(c.z0m), dbl(0.23), dbl(1./3.))
        O.orc_boundary_cyclic(Gh, ptr(want), cm.EDGE_BOTH)
        d = B.DevCase(be, c); f = d.fields()
        p = capi.MhhDiffParams(); p.cs = 0.23; p.tPr = 1./3.; p.surface_model = sm; p.grav = 9.81
        dth = be.arr(thref); p.thref = be.ptr(dth).value
        ml = B.mlen0(be, g, 0.23); p.mlen0 = be.ptr(ml).value
        B.ok(be, be.lib.mhh_diff_exec_viscosity(d.G, dif, C.byref(f), C.byref(p), be.stream))
        assert cm.ulp_diff(be.host(d.evisc)[g.kstart:g.kend], want[g.kstart:g.kend]) <= 8, g.shape3


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("limited", [(0,), (1,), (0, 1)])
def test_fused_rhs_with_fluxlimit_list(be, limited, dtype):
    """advec.fluxlimit_list (src/advec_2i5.cxx:39,921): the listed scalars take the Koren-limited scheme inside
    mhh_rhs_exec and mhh_advec_exec + mhh_diff_exec; everything else keeps the 2i5 bits."""
    adv, dif, sm = cm.ADVEC_2I5, cm.DIFF_SMAG2, 1
    for g in grids2(dtype)[:2]:
        c = cm.Case(g, nscalars=2)
        c.u, c.v, c.w, c.s[0] = cm.limiter_inputs(c, dtype)
        want = _oracle_rhs(c, adv, dif, sm, limited=limited)
        p = capi.MhhDiffParams(); p.cs = 0.23; p.tPr = 1./3.; p.surface_model = sm
        for fused in (False, True):
            d = B.DevCase(be, c); f = d.fields()
            for n in limited:
                f.s_fluxlimit[n] = 1
            if fused:
                B.ok(be, be.lib.mhh_rhs_exec(d.G, adv, dif, C.byref(f), C.byref(p), be.stream))
            else:
                B.ok(be, be.lib.mhh_advec_exec(d.G, adv, C.byref(f), be.stream))
                B.ok(be, be.lib.mhh_diff_exec(d.G, dif, C.byref(f), C.byref(p), be.stream))
            got = (be.host(d.ut), be.host(d.vt), be.host(d.wt), [be.host(x) for x in d.st])
            for a, b, nm in zip(got[:3], want[:3], "uvw"):
                assert same(a, b), (fused, nm, g.shape3, cm.ulp_diff(a, b))
            for n, (a, b) in enumerate(zip(got[3], want[3])):
                assert same(a, b), (fused, "s%d" % n, limited, g.shape3, cm.ulp_diff(a, b))
        f = d.fields(); f.s_fluxlimit[0] = 1
        assert be.lib.mhh_rhs_exec(d.G, cm.ADVEC_2, cm.DIFF_2, C.byref(f), C.byref(p), be.stream) != 0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("adv,dif,order", [(cm.ADVEC_2, cm.DIFF_2, 2), (cm.ADVEC_2I5, cm.DIFF_SMAG2, 2), (cm.ADVEC_4, cm.DIFF_4, 4)])
def test_dry_buoyancy_standalone_and_folded(be, adv, dif, order, dtype):
    """Thermo_dry buoyancy tendency (src/thermo_dry.cxx:165-197) as its own kernel and folded into the fused RHS, where it
    is the first term added to wt like in Model::exec."""
    O = cm.oracle(); grav = 9.81
    for g in (grids4(dtype) if adv == cm.ADVEC_4 else grids2(dtype))[:2]:
        c = cm.Case(g, nscalars=2); Gh = g.host_struct()
        threfh = (300. + 0.37*np.arange(g.kcells)).astype(dtype)
        d = B.DevCase(be, c); dth = be.arr(threfh)
        want = c.wt.copy(); O.orc_buoyancy_tend(Gh, order, ptr(want), ptr(c.s[0]), ptr(threfh), dbl(grav))
        t = be.arr(c.wt)
        B.ok(be, be.lib.mhh_thermo_dry_buoyancy_tend(d.G, order, be.ptr(t), be.ptr(d.s[0]), be.ptr(dth), grav, be.stream))
        assert same(be.host(t), want) and not np.array_equal(want, c.wt)
        for nth in (0, 1):          # scalar 0 rides inside the march kernel, scalar 1 takes the separate kernel first
            want = _oracle_rhs(c, adv, dif, 1 if dif == cm.DIFF_SMAG2 else 0, buoy=(order, nth, threfh, grav))
            d = B.DevCase(be, c); f = d.fields()
            p = capi.MhhDiffParams(); p.cs = 0.23; p.tPr = 1./3.; p.surface_model = 1 if dif == cm.DIFF_SMAG2 else 0
            p.buoyancy = order; p.th_for_N2 = nth; p.threfh = be.ptr(dth).value; p.grav = grav
            B.ok(be, be.lib.mhh_rhs_exec(d.G, adv, dif, C.byref(f), C.byref(p), be.stream))
            got = (be.host(d.ut), be.host(d.vt), be.host(d.wt), [be.host(x) for x in d.st])
            for a, b, nm in zip(got[:3], want[:3], "uvw"):
                assert same(a, b), ("folded buoyancy", nm, adv, nth, g.shape3, cm.ulp_diff(a, b))
            for a, b in zip(got[3], want[3]):
                assert same(a, b)
        p.th_for_N2 = 7
        assert be.lib.mhh_rhs_exec(d.G, adv, dif, C.byref(f), C.byref(p), be.stream) != 0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("sm,neutral", [(1, 0), (0, 0), (1, 1)])
def test_exec_viscosity(be, sm, neutral, dtype):
    """Diff_smag2::exec_viscosity as one pass (strain2 + inline N2 + evisc + cyclic) vs the oracle's three steps."""
    O = cm.oracle()
    for g in grids2(dtype)[:2]:
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
        d = B.DevCase(be, c); f = d.fields()
        p = capi.MhhDiffParams(); p.cs = cs; p.tPr = tPr; p.surface_model = sm; p.neutral = neutral
        p.N2 = None; p.th_for_N2 = 0; dthref = be.arr(thref); p.thref = be.ptr(dthref).value; p.grav = grav
        ml = B.mlen0(be, g, cs); p.mlen0 = be.ptr(ml).value
        B.ok(be, be.lib.mhh_diff_exec_viscosity(d.G, cm.DIFF_SMAG2, C.byref(f), C.byref(p), be.stream))
        ev = be.host(d.evisc)
        k0, k1 = (g.kstart, g.kend) if sm else (g.kstart-1, g.kend+1)
        assert cm.ulp_diff(ev[k0:k1], want[k0:k1]) <= 8, cm.ulp_diff(ev[k0:k1], want[k0:k1])


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("order", [2, 4])
def test_pres_exec_callback_fused_equals_staged(be, order, dtype):
    """mhh_pres_exec with Pres::input as the forward transform's load callback and the unpack as the inverse transform's
    store callback, against input -> solve -> output as three stages: the same transforms, the same bits (p with all its
    ghost cells, and the corrected tendencies)."""
    O = cm.oracle()
    gl = [cm.grid_2nd(16, 12, 10, gc=(1, 1, 1), dtype=dtype), cm.grid_2nd(12, 10, 8, gc=(3, 3, 1), dtype=dtype), cm.grid_2nd(12, 1, 8, gc=(1, 1, 1), dtype=dtype),
          cm.grid_2nd(4, 3, 6, gc=(3, 3, 1), dtype=dtype)] if order == 2 else [cm.grid_4th(16, 12, 12, dtype=dtype), cm.grid_4th(12, 1, 8, dtype=dtype)]
    for g in gl:
        c = cm.Case(g, rho=("random" if order == 2 else "one"), periodic=True)
        Gh = g.host_struct(); dt = 0.7
        out = {}
        for form in ("fused", "staged"):
            d = B.DevCase(be, c); f = d.fields()
            plan = capi.PLAN()
            B.ok(be, be.lib.mhh_pres_plan_create(Gh, order, ptr(g.dz), ptr(g.dzhi), ptr(g.dzi4), ptr(g.dzhi4), ptr(c.rhoref), ptr(c.rhorefh), C.byref(plan)))
            os.environ["MHH_PRES_FUSED"] = "1" if form == "fused" else "0"
            os.environ["MHH_PRES_LDS"] = "0"                     # both are forms of the rocFFT path
            try:
                B.ok(be, be.lib.mhh_pres_exec(plan, d.G, C.byref(f), dt, be.stream))
            finally:
                os.environ.pop("MHH_PRES_FUSED", None); os.environ.pop("MHH_PRES_LDS", None)
            out[form] = [be.host(x) for x in (d.p, d.ut, d.vt, d.wt)]
            be.lib.mhh_pres_plan_destroy(plan)
        for a, b, nm in zip(out["fused"], out["staged"], ("p", "ut", "vt", "wt")):
            assert same(a, b), (order, g.shape3, nm, cm.ulp_diff(a, b))
        assert not np.array_equal(out["fused"][0], c.p)


@pytest.mark.parametrize("dtype", DTYPES)
def test_unpack_normalisation_is_two_divisions_bit_for_bit(be, dtype):
    """The normalisation after the inverse transform is value / jtot / itot (src/fft.cxx); with power-of-two extents the
    library multiplies by the exact reciprocals instead -- the bits must be those of the divisions, on values across
    the whole exponent range. An identity 'solve' is not available, so this drives the spectral path with a packed
    field whose transform pair returns itot*jtot*x up to rounding and checks the unpacked p against the division of the
    very same packed solution, read back through mhh_pres_solve's in-place buffer."""
    for shape in ((16, 8, 6), (16, 12, 6), (8, 1, 6)):
        g = cm.grid_2nd(*shape, gc=(2, 2, 1), dtype=dtype)
        c = cm.Case(g, rho="random", periodic=True); Gh = g.host_struct()
        d = B.DevCase(be, c); f = d.fields()
        plan = capi.PLAN()
        B.ok(be, be.lib.mhh_pres_plan_create(Gh, 2, ptr(g.dz), ptr(g.dzhi), ptr(g.dzi4), ptr(g.dzhi4), ptr(c.rhoref), ptr(c.rhorefh), C.byref(plan)))
        rng = np.random.default_rng(5)
        pk0 = (rng.standard_normal((g.ktot, g.jtot, g.itot)) * 10.0**rng.integers(-30, 30, (g.ktot, g.jtot, g.itot))).astype(dtype)
        pk = be.arr(pk0)
        B.ok(be, be.lib.mhh_pres_solve(plan, d.G, C.byref(f), be.ptr(pk), be.stream))
        sol = be.host(pk)                                   # the un-normalised solution the unpack read
        want = (sol / dtype(g.jtot) / dtype(g.itot)).astype(dtype)
        got = be.host(d.p)[g.kstart:g.kend, g.jstart:g.jend, g.istart:g.iend]
        assert same(got, want), (shape, cm.ulp_diff(got, want))
        be.lib.mhh_pres_plan_destroy(plan)


@pytest.mark.parametrize("dtype", DTYPES)
def test_pres4_exec_unpack_and_output_in_one_kernel_equals_two(be, dtype):
    """The same for pres_4: four mirrored ghost levels of p, 4-point gradients gathered from the packed solution (wrapped in the
    horizontal, mirrored across the walls), wt untouched at kstart, vt untouched on a 2-D grid."""
    for g in [cm.grid_4th(16, 12, 12, dtype=dtype), cm.grid_4th(12, 1, 8, dtype=dtype), cm.grid_4th(16, 8, 8, dtype=dtype), cm.grid_4th(6, 5, 4, dtype=dtype)]:
        c = cm.Case(g, rho="one", periodic=True)
        for m in (1, 2):
            c.w[g.kstart-m] = -c.w[g.kstart+m]; c.w[g.kend+m] = -c.w[g.kend-m]
        Gh = g.host_struct(); dt = 0.7
        out = {}
        for form in ("one", "two"):
            d = B.DevCase(be, c); f = d.fields()
            plan = capi.PLAN()
            B.ok(be, be.lib.mhh_pres_plan_create(Gh, 4, ptr(g.dz), ptr(g.dzhi), ptr(g.dzi4), ptr(g.dzhi4), ptr(c.rhoref), ptr(c.rhorefh), C.byref(plan)))
            if form == "two":
                os.environ["MHH_PRES_UNPACK_OUT"] = "0"
            try:
                B.ok(be, be.lib.mhh_pres_exec(plan, d.G, C.byref(f), dt, be.stream))
            finally:
                os.environ.pop("MHH_PRES_UNPACK_OUT", None)
            out[form] = [be.host(x) for x in (d.p, d.ut, d.vt, d.wt)]
            be.lib.mhh_pres_plan_destroy(plan)
        for x, y, nm in zip(out["one"], out["two"], ("p", "ut", "vt", "wt")):
            assert same(x, y), (g.shape3, nm, cm.ulp_diff(x, y))
        assert not np.array_equal(out["one"][1], c.ut)


@pytest.mark.parametrize("dtype", DTYPES)
def test_pres2_exec_unpack_and_output_in_one_kernel_equals_two(be, dtype):
    """mhh_pres_exec (order 2) unpacks the solution and applies Pres_2::output in one kernel; MHH_PRES_UNPACK_OUT=0 runs
    them as the two kernels of mhh_pres_solve + mhh_pres_output. Same bits: p with every ghost cell, ut, vt, wt."""
    gl = [cm.grid_2nd(16, 12, 10, gc=(1, 1, 1), dtype=dtype), cm.grid_2nd(12, 10, 8, gc=(3, 3, 2), dtype=dtype), cm.grid_2nd(12, 1, 8, gc=(1, 1, 1), dtype=dtype),
          cm.grid_2nd(4, 3, 6, gc=(3, 3, 1), dtype=dtype), cm.grid_2nd(300, 5, 4, gc=(2, 2, 1), dtype=dtype), cm.grid_2nd(16, 8, 6, gc=(2, 2, 1), dtype=dtype),
          cm.grid_2nd(8, 1, 6, gc=(1, 1, 1), dtype=dtype)]
    for g in gl:
        c = cm.Case(g, rho="random", periodic=True)
        Gh = g.host_struct(); dt = 0.7
        out = {}
        for form in ("one", "two"):
            d = B.DevCase(be, c); f = d.fields()
            plan = capi.PLAN()
            B.ok(be, be.lib.mhh_pres_plan_create(Gh, 2, ptr(g.dz), ptr(g.dzhi), ptr(g.dzi4), ptr(g.dzhi4), ptr(c.rhoref), ptr(c.rhorefh), C.byref(plan)))
            if form == "two":
                os.environ["MHH_PRES_UNPACK_OUT"] = "0"
            os.environ["MHH_PRES_LDS"] = "0"                     # both are forms of the rocFFT path
            try:
                B.ok(be, be.lib.mhh_pres_exec(plan, d.G, C.byref(f), dt, be.stream))
            finally:
                os.environ.pop("MHH_PRES_UNPACK_OUT", None); os.environ.pop("MHH_PRES_LDS", None)
            out[form] = [be.host(x) for x in (d.p, d.ut, d.vt, d.wt)]
            be.lib.mhh_pres_plan_destroy(plan)
        for x, y, nm in zip(out["one"], out["two"], ("p", "ut", "vt", "wt")):
            assert same(x, y), (g.shape3, nm, cm.ulp_diff(x, y))
        assert not np.array_equal(out["one"][1], c.ut)


@pytest.mark.parametrize("dtype", DTYPES)
def test_pres2_lds_transform_form(be, dtype):
    """Pres_2::exec as three kernels with the transforms in LDS (csrc/pres_lds.h; power-of-two itot, jtot -- the form
    mhh_pres_exec takes by default there). Stage 1 against numpy's real transform of the oracle's Pres_2::input (the modes
    kx = 0 and kx = itot/2 share column 0); the whole operator against the oracle's input -> solve -> output: p with all its
    ghost cells and the three corrected tendencies within the pressure tolerance; levels-per-block that do and do not divide
    kmax; the ghost-cell side effects of Pres_2::input on ut, vt bit for bit."""
    O = cm.oracle()
    tol = 1e-11 if dtype == np.float64 else 2e-4
    cases = [((16, 8, 6), (2, 2, 1), None), ((32, 16, 10), (3, 3, 1), "3"), ((64, 8, 9), (1, 1, 1), "4"), ((16, 64, 17), (3, 3, 1), None), ((128, 32, 5), (3, 3, 2), "2")]
    for shape, gc, kc in cases:
        g = cm.grid_2nd(*shape, gc=gc, dtype=dtype)
        c = cm.Case(g, rho="random", periodic=True); Gh = g.host_struct(); dt = 0.7
        d = B.DevCase(be, c); f = d.fields()
        plan = capi.PLAN()
        B.ok(be, be.lib.mhh_pres_plan_create(Gh, 2, ptr(g.dz), ptr(g.dzhi), ptr(g.dzi4), ptr(g.dzhi4), ptr(c.rhoref), ptr(c.rhorefh), C.byref(plan)))
        if kc: os.environ["MHH_PRES_LDS_KC"] = kc
        try:
            assert be.lib.mhh_pres_plan_has_lds_form(plan) == 1
            assert be.lib.mhh_pres_exec_form(plan) == 0                  # small grid: mhh_pres_exec stays with the staged form ...
            os.environ["MHH_PRES_LDS"] = "1"
            assert be.lib.mhh_pres_exec_form(plan) == 1                  # ... unless told otherwise
            os.environ["MHH_PRES_LDS"] = "0"
            assert be.lib.mhh_pres_exec_form(plan) == 0
            os.environ.pop("MHH_PRES_LDS")
            pk_want = np.zeros((g.ktot, g.jtot, g.itot), dtype=dtype)
            ut, vt, wt = c.ut.copy(), c.vt.copy(), c.wt.copy()
            O.orc_pres_input(Gh, 2, ptr(pk_want), ptr(c.u), ptr(c.v), ptr(c.w), ptr(ut), ptr(vt), ptr(wt), ptr(c.rhoref), ptr(c.rhorefh), dbl(dt))
            B.ok(be, be.lib.mhh_pres_lds_stage(plan, d.G, C.byref(f), dt, 1, be.stream)); be.sync()
            assert same(be.host(d.ut), ut) and same(be.host(d.vt), vt)
            nh = g.itot // 2
            spec = be.host(be.view(be.lib.mhh_pres_plan_spectral(plan), (g.ktot, nh, g.jtot, 2), dtype))
            want = np.fft.rfft(pk_want.astype(np.float64), axis=2).transpose(0, 2, 1)
            got = spec[..., 0] + 1j*spec[..., 1]
            scale = np.abs(want).max()
            assert np.abs(got[:, 1:] - want[:, 1:nh]).max() <= tol*scale
            assert np.abs(got[:, 0].real - want[:, 0].real).max() <= tol*scale and np.abs(got[:, 0].imag - want[:, nh].real).max() <= tol*scale
            B.ok(be, be.lib.mhh_pres_lds_stage(plan, d.G, C.byref(f), dt, 2, be.stream))
            B.ok(be, be.lib.mhh_pres_lds_stage(plan, d.G, C.byref(f), dt, 3, be.stream))
            p_want = np.zeros(g.shape3, dtype=dtype); pk_tmp = pk_want.copy()
            O.orc_pres_solve(Gh, 2, ptr(p_want), ptr(pk_tmp), ptr(c.rhoref), ptr(c.rhorefh))
            O.orc_pres_output(Gh, 2, ptr(ut), ptr(vt), ptr(wt), ptr(p_want))
            sl = (slice(g.kstart-1, g.kend), slice(None), slice(None))
            pscale = np.abs(p_want).max()
            assert np.abs(be.host(d.p)[sl] - p_want[sl]).max() <= tol*pscale, (shape, np.abs(be.host(d.p)[sl] - p_want[sl]).max()/pscale)
            for got_t, want_t, nm in ((d.ut, ut, "ut"), (d.vt, vt, "vt"), (d.wt, wt, "wt")):
                assert np.abs(be.host(got_t) - want_t).max() <= tol*max(np.abs(want_t).max(), pscale/float(min(g.dx, g.dy))), (shape, nm)
            # mhh_pres_exec in this form (by itself only on large grids): same bits as the three stages
            d2 = B.DevCase(be, c); f2 = d2.fields()
            os.environ["MHH_PRES_LDS"] = "1"
            try:
                B.ok(be, be.lib.mhh_pres_exec(plan, d2.G, C.byref(f2), dt, be.stream))
            finally:
                os.environ.pop("MHH_PRES_LDS", None)
            for x, y in ((d.p, d2.p), (d.ut, d2.ut), (d.vt, d2.vt), (d.wt, d2.wt)):
                assert same(be.host(x), be.host(y))
            # and it is a projection: nothing left for a second solve
            pk = be.zeros((g.ktot, g.jtot, g.itot), dtype)
            B.ok(be, be.lib.mhh_pres_input(plan, d2.G, C.byref(f2), dt, be.ptr(pk), be.stream))
            assert np.abs(be.host(pk)).max() <= (1e-9 if dtype == np.float64 else 2e-2) * np.abs(pk_want).max()
        finally:
            os.environ.pop("MHH_PRES_LDS_KC", None)
            be.lib.mhh_pres_plan_destroy(plan)
    # grids the form does not cover keep the staged one
    g = cm.grid_2nd(12, 10, 8, gc=(3, 3, 1), dtype=dtype); c = cm.Case(g, rho="random", periodic=True)
    plan = capi.PLAN()
    B.ok(be, be.lib.mhh_pres_plan_create(g.host_struct(), 2, ptr(g.dz), ptr(g.dzhi), ptr(g.dzi4), ptr(g.dzhi4), ptr(c.rhoref), ptr(c.rhorefh), C.byref(plan)))
    assert be.lib.mhh_pres_plan_has_lds_form(plan) == 0
    be.lib.mhh_pres_plan_destroy(plan)


def _lds4_form_against_oracle(be, shape, dtype, tol, kc=None, stages=False):
    """Pres_4::exec in the LDS-transform form (csrc/pres_lds4.h) on one grid against the oracle's input -> solve -> output: p with
    its periodic halo and its four mirrored ghost levels, the three corrected tendencies, the ghost-cell side effects of
    Pres_4::input; stages=True also runs the three stages one by one (same bits) and checks stage 1 against numpy's transform."""
    O = cm.oracle()
    g = cm.grid_4th(*shape, dtype=dtype)
    c = cm.Case(g, rho="one", periodic=True)
    for m in (1, 2):
        c.w[g.kstart-m] = -c.w[g.kstart+m]; c.w[g.kend+m] = -c.w[g.kend-m]
    Gh = g.host_struct(); dt = 0.7
    d = B.DevCase(be, c); f = d.fields()
    plan = capi.PLAN()
    B.ok(be, be.lib.mhh_pres_plan_create(Gh, 4, ptr(g.dz), ptr(g.dzhi), ptr(g.dzi4), ptr(g.dzhi4), ptr(c.rhoref), ptr(c.rhorefh), C.byref(plan)))
    if kc: os.environ["MHH_PRES_LDS_KC"] = kc
    try:
        assert be.lib.mhh_pres_plan_has_lds_form(plan) == 1, shape
        os.environ["MHH_PRES_LDS"] = "1"
        try:
            assert be.lib.mhh_pres_exec_form(plan) == 1
            B.ok(be, be.lib.mhh_pres_exec(plan, d.G, C.byref(f), dt, be.stream))
        finally:
            os.environ.pop("MHH_PRES_LDS", None)
        pk = np.zeros((g.ktot, g.jtot, g.itot), dtype=dtype); p_want = np.zeros(g.shape3, dtype=dtype)
        ut, vt, wt = c.ut.copy(), c.vt.copy(), c.wt.copy()
        O.orc_pres_input(Gh, 4, ptr(pk), ptr(c.u), ptr(c.v), ptr(c.w), ptr(ut), ptr(vt), ptr(wt), ptr(c.rhoref), ptr(c.rhorefh), dbl(dt))
        pk_in = pk.copy()
        O.orc_pres_solve(Gh, 4, ptr(p_want), ptr(pk), ptr(c.rhoref), ptr(c.rhorefh))
        O.orc_pres_output(Gh, 4, ptr(ut), ptr(vt), ptr(wt), ptr(p_want))
        sl = (slice(g.kstart-2, g.kend+2), slice(None), slice(None))
        pscale = np.abs(p_want).max()
        err = np.abs(be.host(d.p)[sl] - p_want[sl]).max() / pscale
        assert err <= tol, (shape, err)
        for got_t, want_t, nm in ((d.ut, ut, "ut"), (d.vt, vt, "vt"), (d.wt, wt, "wt")):
            terr = np.abs(be.host(got_t) - want_t).max() / max(np.abs(want_t).max(), pscale/float(min(g.dx, g.dy)))
            assert terr <= 10*tol, (shape, nm, terr)
        if stages:
            d2 = B.DevCase(be, c); f2 = d2.fields()
            B.ok(be, be.lib.mhh_pres_lds_stage(plan, d2.G, C.byref(f2), dt, 1, be.stream)); be.sync()
            nh = g.itot // 2
            spec = be.host(be.view(be.lib.mhh_pres_plan_spectral(plan), (g.ktot, nh, g.jtot, 2), dtype))
            want = np.fft.rfft(pk_in.astype(np.float64), axis=2).transpose(0, 2, 1)
            got = spec[..., 0] + 1j*spec[..., 1]
            scale = np.abs(want).max()
            assert np.abs(got[:, 1:] - want[:, 1:nh]).max() <= tol*scale
            assert np.abs(got[:, 0].real - want[:, 0].real).max() <= tol*scale and np.abs(got[:, 0].imag - want[:, nh].real).max() <= tol*scale
            B.ok(be, be.lib.mhh_pres_lds_stage(plan, d2.G, C.byref(f2), dt, 2, be.stream))
            B.ok(be, be.lib.mhh_pres_lds_stage(plan, d2.G, C.byref(f2), dt, 3, be.stream))
            for x, y in ((d.p, d2.p), (d.ut, d2.ut), (d.vt, d2.vt), (d.wt, d2.wt)):
                assert same(be.host(x), be.host(y))
            # a projection: nothing left for a second solve
            pk2 = be.zeros((g.ktot, g.jtot, g.itot), dtype)
            B.ok(be, be.lib.mhh_pres_input(plan, d2.G, C.byref(f2), dt, be.ptr(pk2), be.stream))
            assert np.abs(be.host(pk2)).max() <= (1e-9 if dtype == np.float64 else 2e-2) * np.abs(pk_in).max()
    finally:
        os.environ.pop("MHH_PRES_LDS_KC", None)
        be.lib.mhh_pres_plan_destroy(plan)


@pytest.mark.parametrize("dtype", DTYPES)
def test_pres4_lds_transform_form(be, dtype):
    """Pres_4::exec as three kernels with the transforms in LDS + the wt pass (csrc/pres_lds4.h): run-time-size and compile-time-size
    instantiations, levels-per-block that do and do not divide kmax, kmax not a multiple of the eight levels of a round."""
    tol = 1e-11 if dtype == np.float64 else 2e-4
    for shape, kc in (((16, 8, 8), None), ((32, 16, 13), "3"), ((64, 8, 9), "4"), ((16, 64, 17), None), ((128, 32, 6), "2")):
        _lds4_form_against_oracle(be, shape, dtype, tol, kc=kc, stages=True)
    # grids the form does not cover keep the staged one
    g = cm.grid_4th(12, 10, 8, dtype=dtype); c = cm.Case(g, rho="one", periodic=True)
    plan = capi.PLAN()
    B.ok(be, be.lib.mhh_pres_plan_create(g.host_struct(), 4, ptr(g.dz), ptr(g.dzhi), ptr(g.dzi4), ptr(g.dzhi4), ptr(c.rhoref), ptr(c.rhorefh), C.byref(plan)))
    assert be.lib.mhh_pres_plan_has_lds_form(plan) == 0
    be.lib.mhh_pres_plan_destroy(plan)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", DTYPES)
def test_pres4_lds_form_at_instantiated_row_lengths(dtype):
    """The compile-time-size instantiations of the pres_4 kernels against the oracle, moser600's among them (512 x 256, fp64)."""
    be = B.get("hip")
    tol = 1e-11 if dtype == np.float64 else 2e-4
    for shape in [(512, 256, 12), (256, 128, 9), (128, 64, 10), (1024, 512, 8), (256, 512, 8)]:
        _lds4_form_against_oracle(be, shape, dtype, tol)


def _lds_form_against_oracle(be, shape, gc, dtype, tol):
    """Pres_2::exec in the LDS-transform form on one grid against the oracle's input -> solve -> output."""
    O = cm.oracle()
    g = cm.grid_2nd(*shape, gc=gc, dtype=dtype)
    c = cm.Case(g, rho="random", periodic=True); Gh = g.host_struct(); dt = 0.7
    d = B.DevCase(be, c); f = d.fields()
    plan = capi.PLAN()
    B.ok(be, be.lib.mhh_pres_plan_create(Gh, 2, ptr(g.dz), ptr(g.dzhi), ptr(g.dzi4), ptr(g.dzhi4), ptr(c.rhoref), ptr(c.rhorefh), C.byref(plan)))
    try:
        assert be.lib.mhh_pres_plan_has_lds_form(plan) == 1, shape
        os.environ["MHH_PRES_LDS"] = "1"
        try:
            B.ok(be, be.lib.mhh_pres_exec(plan, d.G, C.byref(f), dt, be.stream))
        finally:
            os.environ.pop("MHH_PRES_LDS", None)
        pk = np.zeros((g.ktot, g.jtot, g.itot), dtype=dtype); p_want = np.zeros(g.shape3, dtype=dtype)
        ut, vt, wt = c.ut.copy(), c.vt.copy(), c.wt.copy()
        O.orc_pres_input(Gh, 2, ptr(pk), ptr(c.u), ptr(c.v), ptr(c.w), ptr(ut), ptr(vt), ptr(wt), ptr(c.rhoref), ptr(c.rhorefh), dbl(dt))
        O.orc_pres_solve(Gh, 2, ptr(p_want), ptr(pk), ptr(c.rhoref), ptr(c.rhorefh))
        O.orc_pres_output(Gh, 2, ptr(ut), ptr(vt), ptr(wt), ptr(p_want))
        sl = (slice(g.kstart-1, g.kend), slice(None), slice(None))
        pscale = np.abs(p_want).max()
        err = np.abs(be.host(d.p)[sl] - p_want[sl]).max() / pscale
        assert err <= tol, (shape, err)
        for got_t, want_t, nm in ((d.ut, ut, "ut"), (d.vt, vt, "vt"), (d.wt, wt, "wt")):
            assert np.abs(be.host(got_t) - want_t).max() <= tol*max(np.abs(want_t).max(), pscale/float(min(g.dx, g.dy))), (shape, nm)
    finally:
        be.lib.mhh_pres_plan_destroy(plan)


@pytest.mark.parametrize("dtype", DTYPES)
def test_pres2_lds_y_stage_with_two_blocks_per_column(be, dtype):
    """The y stage of the LDS form as a twisted factorisation (csrc/pres_lds.h, 2t: the lower levels eliminated bottom-up by one block,
    the upper ones top-down by a second, a 2 x 2 system where they meet) against the oracle, like the one-block form: upper halves of
    one level, of whole and of partial rounds of eight; and against the one-block form within the same tolerance."""
    tol = 1e-11 if dtype == np.float64 else 2e-4
    os.environ["MHH_PRES_Y_TWISTED"] = "1"
    try:
        for shape in ((16, 64, 17), (32, 16, 24), (16, 8, 37), (64, 32, 40), (128, 32, 16)):
            _lds_form_against_oracle(be, shape, (3, 3, 1), dtype, tol)
    finally:
        os.environ.pop("MHH_PRES_Y_TWISTED", None)
    # the switch does something: the two forms differ in rounding (and only in rounding)
    g = cm.grid_2nd(32, 16, 24, gc=(3, 3, 1), dtype=dtype); c = cm.Case(g, rho="random", periodic=True)
    plan = capi.PLAN()
    B.ok(be, be.lib.mhh_pres_plan_create(g.host_struct(), 2, ptr(g.dz), ptr(g.dzhi), ptr(g.dzi4), ptr(g.dzhi4), ptr(c.rhoref), ptr(c.rhorefh), C.byref(plan)))
    out = {}
    try:
        for tw in ("0", "1"):
            d = B.DevCase(be, c); f = d.fields()
            os.environ["MHH_PRES_LDS"] = "1"; os.environ["MHH_PRES_Y_TWISTED"] = tw
            try:
                B.ok(be, be.lib.mhh_pres_exec(plan, d.G, C.byref(f), 0.7, be.stream))
            finally:
                os.environ.pop("MHH_PRES_LDS", None); os.environ.pop("MHH_PRES_Y_TWISTED", None)
            out[tw] = be.host(d.p)
    finally:
        be.lib.mhh_pres_plan_destroy(plan)
    assert not np.array_equal(out["0"], out["1"])
    assert np.abs(out["0"] - out["1"]).max() <= tol * np.abs(out["0"]).max()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", DTYPES)
def test_pres2_lds_form_at_every_instantiated_row_length(dtype):
    """The compile-time-size instantiations of the three kernels (itot = 128 ... 1024, jtot = 64 ... 1024; csrc/k_pres.hip) each
    against the ORACLE directly, the benchmark's among them: 512 x 512 (fp64, drycblles 512^3) and 1024 x 512 / 1024 x 1024 (fp32,
    gabls1). A few levels suffice: the transforms are per level, the vertical solve is covered by the small cases. GPU only: a
    1024-thread block is a thousand fibres per block on the emulation."""
    be = B.get("hip")
    tol = 1e-11 if dtype == np.float64 else 2e-4
    shapes = [(512, 512, 6), (256, 256, 5), (128, 64, 7), (1024, 128, 4), (256, 512, 4), (512, 256, 9)]
    shapes += [(1024, 512, 4), (128, 1024, 3), (1024, 1024, 3)] if dtype == np.float32 else [(1024, 512, 3)]
    for shape in shapes:
        _lds_form_against_oracle(be, shape, (3, 3, 1), dtype, tol)
    if dtype == np.float64:     # fp64 rows of 1024 along y have no instantiation (scratch): the plan says so and mhh_pres_exec takes the staged form
        g = cm.grid_2nd(128, 1024, 3, gc=(3, 3, 1), dtype=dtype); c = cm.Case(g, rho="random", periodic=True)
        plan = capi.PLAN()
        B.ok(be, be.lib.mhh_pres_plan_create(g.host_struct(), 2, ptr(g.dz), ptr(g.dzhi), ptr(g.dzi4), ptr(g.dzhi4), ptr(c.rhoref), ptr(c.rhorefh), C.byref(plan)))
        assert be.lib.mhh_pres_plan_has_lds_form(plan) == 0 and be.lib.mhh_pres_exec_form(plan) == 0
        be.lib.mhh_pres_plan_destroy(plan)


@pytest.mark.parametrize("dtype", DTYPES)
def test_pres2_lds_plans_of_different_sizes_do_not_disturb_each_other(be, dtype):
    """The dynamic-LDS ceiling is a property of the kernel, not of the plan: a large plan, then a small plan of the same
    instantiation family, then the large plan again (the small plan used to lower the ceiling under the large one)."""
    tol = 1e-11 if dtype == np.float64 else 2e-4
    O = cm.oracle()
    gl, gs = cm.grid_2nd(64, 32, 6, gc=(3, 3, 1), dtype=dtype), cm.grid_2nd(16, 8, 6, gc=(3, 3, 1), dtype=dtype)
    cl, cs = cm.Case(gl, rho="random", periodic=True), cm.Case(gs, rho="random", periodic=True)
    plans = []
    for g, c in ((gl, cl), (gs, cs)):
        plan = capi.PLAN()
        B.ok(be, be.lib.mhh_pres_plan_create(g.host_struct(), 2, ptr(g.dz), ptr(g.dzhi), ptr(g.dzi4), ptr(g.dzhi4), ptr(c.rhoref), ptr(c.rhorefh), C.byref(plan)))
        plans.append(plan)
    os.environ["MHH_PRES_LDS"] = "1"
    try:
        for g, c, plan in ((gl, cl, plans[0]), (gs, cs, plans[1]), (gl, cl, plans[0])):
            d = B.DevCase(be, c); f = d.fields(); dt = 0.7
            B.ok(be, be.lib.mhh_pres_exec(plan, d.G, C.byref(f), dt, be.stream))
            Gh = g.host_struct()
            pk = np.zeros((g.ktot, g.jtot, g.itot), dtype=dtype); p_want = np.zeros(g.shape3, dtype=dtype)
            ut, vt, wt = c.ut.copy(), c.vt.copy(), c.wt.copy()
            O.orc_pres_input(Gh, 2, ptr(pk), ptr(c.u), ptr(c.v), ptr(c.w), ptr(ut), ptr(vt), ptr(wt), ptr(c.rhoref), ptr(c.rhorefh), dbl(dt))
            O.orc_pres_solve(Gh, 2, ptr(p_want), ptr(pk), ptr(c.rhoref), ptr(c.rhorefh))
            sl = (slice(g.kstart-1, g.kend), slice(None), slice(None))
            assert np.abs(be.host(d.p)[sl] - p_want[sl]).max() <= tol*np.abs(p_want).max(), g.shape3
    finally:
        os.environ.pop("MHH_PRES_LDS", None)
        for plan in plans:
            be.lib.mhh_pres_plan_destroy(plan)


@pytest.mark.parametrize("adv,dif", [(cm.ADVEC_2I4, cm.DIFF_2), (cm.ADVEC_2I62, cm.DIFF_SMAG2), (cm.ADVEC_2I53, cm.DIFF_SMAG2), (cm.ADVEC_4M, cm.DIFF_4)])
def test_rhs_exec_other_scheme_pairs_run_as_two_calls(be, adv, dif):
    """mhh_rhs_exec accepts every pair of valid schemes: pairs without a fused kernel run Advec::exec then Diff::exec."""
    O = cm.oracle()
    g = cm.grid_4th(16, 12, 12) if adv == cm.ADVEC_4M else cm.grid_2nd(16, 12, 10, gc=(3, 3, 2))
    c = cm.Case(g, nscalars=1); Gh = g.host_struct()
    sm = 1 if dif == cm.DIFF_SMAG2 else 0
    ut, vt, wt, st = c.ut.copy(), c.vt.copy(), c.wt.copy(), c.st[0].copy()
    a = (ptr(c.u), ptr(c.v), ptr(c.w), ptr(c.rhoref), ptr(c.rhorefh))
    O.orc_advec_u(Gh, adv, ptr(ut), *a); O.orc_advec_v(Gh, adv, ptr(vt), *a); O.orc_advec_w(Gh, adv, ptr(wt), *a); O.orc_advec_s(Gh, adv, ptr(st), ptr(c.s[0]), *a)
    cpy = cm.Case(g, nscalars=1)
    cpy.ut, cpy.vt, cpy.wt, cpy.st = ut, vt, wt, [st]
    want = None
    if dif == cm.DIFF_SMAG2:
        O.orc_smag2_diff_u(Gh, sm, ptr(ut), ptr(c.u), ptr(c.v), ptr(c.w), ptr(c.evisc), ptr(c.u_fluxbot), ptr(c.u_fluxtop), ptr(c.rhoref), ptr(c.rhorefh), dbl(1e-5))
        O.orc_smag2_diff_v(Gh, sm, ptr(vt), ptr(c.u), ptr(c.v), ptr(c.w), ptr(c.evisc), ptr(c.v_fluxbot), ptr(c.v_fluxtop), ptr(c.rhoref), ptr(c.rhorefh), dbl(1e-5))
        O.orc_smag2_diff_w(Gh, ptr(wt), ptr(c.u), ptr(c.v), ptr(c.w), ptr(c.evisc), ptr(c.rhoref), ptr(c.rhorefh), dbl(1e-5))
        O.orc_smag2_diff_c(Gh, sm, ptr(st), ptr(c.s[0]), ptr(c.evisc), ptr(c.s_fluxbot), ptr(c.s_fluxtop), ptr(c.rhoref), ptr(c.rhorefh), dbl(1./3.), dbl(1e-5))
    else:
        o = 2 if dif == cm.DIFF_2 else 4
        O.orc_diff_c(Gh, o, ptr(ut), ptr(c.u), dbl(1e-5)); O.orc_diff_c(Gh, o, ptr(vt), ptr(c.v), dbl(1e-5)); O.orc_diff_w(Gh, o, ptr(wt), ptr(c.w), dbl(1e-5))
        O.orc_diff_c(Gh, o, ptr(st), ptr(c.s[0]), dbl(1e-5))
    d = B.DevCase(be, c); f = d.fields()
    p = capi.MhhDiffParams(); p.cs = 0.23; p.tPr = 1./3.; p.surface_model = sm
    B.ok(be, be.lib.mhh_rhs_exec(d.G, adv, dif, C.byref(f), C.byref(p), be.stream))
    for got, w_, nm in ((d.ut, ut, "ut"), (d.vt, vt, "vt"), (d.wt, wt, "wt"), (d.st[0], st, "st")):
        assert same(be.host(got), w_), (adv, dif, nm)
    assert be.lib.mhh_rhs_exec(d.G, 7, dif, C.byref(f), C.byref(p), be.stream) != 0


@pytest.mark.parametrize("dtype", DTYPES)
def test_fused_rhs_on_minimal_and_ragged_grids(be, dtype):
    """Smallest legal vertical extent (every level is wall-adjacent: no interior fast path), tiles narrower than a wave,
    a single row, and extents that are not multiples of the 64 x 4 tile: fused passes against the oracle's operator sequence."""
    cases = [(cm.ADVEC_2I5, cm.DIFF_SMAG2, 1, [(8, 6, 6), (5, 3, 7), (66, 5, 9), (130, 3, 6), (128, 5, 9), (256, 3, 6)]),   # rows of whole 128-cell tiles: two fp32 cells per lane
             (cm.ADVEC_4, cm.DIFF_4, 0, [(8, 6, 6), (6, 4, 5), (66, 5, 9)]),
             (cm.ADVEC_2, cm.DIFF_2, 0, [(4, 3, 2), (66, 5, 3)])]
    for adv, dif, sm, shapes in cases:
        for shape in shapes:
            g = cm.grid_4th(*shape, dtype=dtype) if adv == cm.ADVEC_4 else cm.grid_2nd(*shape, gc=((3, 3, 1) if adv == cm.ADVEC_2I5 else (1, 1, 1)), dtype=dtype)
            c = cm.Case(g, nscalars=1)
            want = _oracle_rhs(c, adv, dif, sm)
            p = capi.MhhDiffParams(); p.cs = 0.23; p.tPr = 1./3.; p.surface_model = sm
            d = B.DevCase(be, c); f = d.fields()
            B.ok(be, be.lib.mhh_rhs_exec(d.G, adv, dif, C.byref(f), C.byref(p), be.stream))
            got = (be.host(d.ut), be.host(d.vt), be.host(d.wt), be.host(d.st[0]))
            for a, b, nm in zip(got, want[:3] + (want[3][0],), ("ut", "vt", "wt", "st")):
                assert same(a, b), (adv, shape, nm, cm.ulp_diff(a, b))


@pytest.mark.parametrize("dtype", DTYPES)
def test_advec25_and_diff_smag2_alone_marching_form_equals_per_field_kernels(be, dtype):
    """Advec::exec (2i5) and Diff::exec (smag2) as separate calls run the marching kernel with one operator's terms
    (u, v, w and the first unlimited scalar; further / flux-limited scalars per field): the bits of the per-field cell
    kernels after each of the two calls, on aligned, unaligned, ragged and minimal grids, with and without surface model."""
    for shape, sm, lim in [((70, 10, 12), 1, (0, 0)), ((17, 9, 8), 0, (0, 1)), ((8, 6, 6), 1, (1, 0)), ((130, 3, 6), 1, (0, 0)), ((128, 5, 7), 1, (0, 0))]:
        g = cm.grid_2nd(*shape, gc=(3, 3, 2), dtype=dtype)
        c = cm.Case(g, nscalars=2)
        p = capi.MhhDiffParams(); p.cs = 0.23; p.tPr = 1./3.; p.surface_model = sm
        out = {}
        for impl in ("march", "cell"):
            d = B.DevCase(be, c); f = d.fields()
            f.s_fluxlimit[0], f.s_fluxlimit[1] = lim
            if impl == "cell":
                os.environ["MHH_ADVEC25_IMPL"] = "cell"; os.environ["MHH_DIFF22_IMPL"] = "cell"
            try:
                B.ok(be, be.lib.mhh_advec_exec(d.G, cm.ADVEC_2I5, C.byref(f), be.stream))
                adv = [be.host(x) for x in (d.ut, d.vt, d.wt, d.st[0], d.st[1])]
                B.ok(be, be.lib.mhh_diff_exec(d.G, cm.DIFF_SMAG2, C.byref(f), C.byref(p), be.stream))
                dif = [be.host(x) for x in (d.ut, d.vt, d.wt, d.st[0], d.st[1])]
            finally:
                os.environ.pop("MHH_ADVEC25_IMPL", None); os.environ.pop("MHH_DIFF22_IMPL", None)
            out[impl] = (adv, dif)
        for stage, nm in ((0, "advec"), (1, "diff")):
            for a, b, fld in zip(out["march"][stage], out["cell"][stage], ("ut", "vt", "wt", "st0", "st1")):
                assert same(a, b), (shape, nm, fld, cm.ulp_diff(a, b))
        assert not np.array_equal(out["march"][0][0], c.ut) and not np.array_equal(out["march"][1][0], out["march"][0][0])


@pytest.mark.parametrize("dtype", DTYPES)
def test_advec4_and_diff4_alone_marching_form_equals_per_field_kernels(be, dtype):
    """Advec_4::exec and Diff_4::exec as separate calls: u, v, w through the 4th-order marching kernel with one operator's
    terms, scalars per field -- the bits of the per-field kernels (MHH_RHS44_IMPL=cell) after each call; 3-D and 2-D grids."""
    for shape in [(70, 10, 12), (17, 9, 8), (12, 1, 8), (66, 5, 9)]:
        g = cm.grid_4th(*shape, dtype=dtype)
        c = cm.Case(g, nscalars=1)
        out = {}
        for impl in ("march", "cell"):
            d = B.DevCase(be, c); f = d.fields()
            if impl == "cell":
                os.environ["MHH_RHS44_IMPL"] = "cell"
            try:
                n0 = be.lib.mhh_stat_rhs44_march_launches()
                B.ok(be, be.lib.mhh_advec_exec(d.G, cm.ADVEC_4, C.byref(f), be.stream))
                adv = [be.host(x) for x in (d.ut, d.vt, d.wt, d.st[0])]
                B.ok(be, be.lib.mhh_diff_exec(d.G, cm.DIFF_4, C.byref(f), None, be.stream))
                dif = [be.host(x) for x in (d.ut, d.vt, d.wt, d.st[0])]
                assert be.lib.mhh_stat_rhs44_march_launches() - n0 == (2 if impl == "march" else 0)
            finally:
                os.environ.pop("MHH_RHS44_IMPL", None)
            out[impl] = (adv, dif)
        for stage, nm in ((0, "advec"), (1, "diff")):
            for a, b, fld in zip(out["march"][stage], out["cell"][stage], ("ut", "vt", "wt", "st")):
                assert same(a, b), (shape, nm, fld, cm.ulp_diff(a, b))


def test_grid_beyond_32_bit_cell_indices_is_refused(be):
    """Cell indices are ints like the reference's ijk: a grid whose ghosted size does not fit is an error, not an overflow."""
    g = cm.grid_2nd(16, 12, 10, gc=(1, 1, 1))
    G = g.host_struct()
    G.contents.imax = G.contents.itot = 2046; G.contents.icells = 2048; G.contents.iend = 2047
    G.contents.jmax = G.contents.jtot = 2046; G.contents.jcells = 2048; G.contents.jend = 2047
    G.contents.ijcells = 2048*2048
    G.contents.kmax = G.contents.ktot = 510; G.contents.kcells = 512; G.contents.kend = 511
    rc = be.lib.mhh_boundary_cyclic(G, None, cm.EDGE_BOTH if hasattr(cm, "EDGE_BOTH") else 3, be.stream)
    assert rc != 0 and b"32-bit" in be.lib.mhh_last_error()


@pytest.mark.parametrize("dtype", DTYPES)
def test_rhs25_march_copy_forms_agree(be, dtype):
    """The two plane-copy forms of k_march.hip (16-byte LDS-DMA, 4-byte LDS-DMA) and the cell kernel give the same bits;
    layouts that are not 16-byte aligned take the 4-byte form by themselves."""
    adv, dif = cm.ADVEC_2I5, cm.DIFF_SMAG2
    for shape in [(70, 10, 12), (17, 9, 8)]:
        g = cm.grid_2nd(*shape, gc=(3, 3, 1), dtype=dtype)
        c = cm.Case(g, nscalars=1)
        p = capi.MhhDiffParams(); p.cs = 0.23; p.tPr = 1./3.; p.surface_model = 1
        out = {}
        for form in ("default", "4", "cell"):
            d = B.DevCase(be, c); f = d.fields()
            key, val = ("MHH_RHS25_IMPL", "cell") if form == "cell" else ("MHH_MARCH_DMA", form)
            if form != "default":
                os.environ[key] = val
            if form == "cell":
                os.environ["MHH_ADVEC25_IMPL"] = "cell"; os.environ["MHH_DIFF22_IMPL"] = "cell"
            try:
                if form == "cell":
                    B.ok(be, be.lib.mhh_advec_exec(d.G, adv, C.byref(f), be.stream))
                    B.ok(be, be.lib.mhh_diff_exec(d.G, dif, C.byref(f), C.byref(p), be.stream))
                else:
                    B.ok(be, be.lib.mhh_rhs_exec(d.G, adv, dif, C.byref(f), C.byref(p), be.stream))
            finally:
                os.environ.pop(key, None); os.environ.pop("MHH_ADVEC25_IMPL", None); os.environ.pop("MHH_DIFF22_IMPL", None)
            out[form] = [be.host(x) for x in (d.ut, d.vt, d.wt, d.st[0])]
        for form in ("4", "cell"):
            for a, b, nm in zip(out["default"], out[form], ("ut", "vt", "wt", "st")):
                assert same(a, b), (shape, form, nm, cm.ulp_diff(a, b))


@pytest.mark.parametrize("rho", ["one", "random"])
@pytest.mark.parametrize("dtype", DTYPES)
def test_rhs25_march_tall_columns(be, dtype, rho):
    """Columns tall enough for the rotated six-level groups of k_march.hip, their remainders and several k-chunks (128
    levels each), Boussinesq (rho == 1: rotated windows) and anelastic (shifted windows) base states, with and without a
    scalar, ragged in i and j: the fused pass equals the oracle's operator-by-operator result bit for bit."""
    adv, dif = cm.ADVEC_2I5, cm.DIFF_SMAG2
    O = cm.oracle()
    for shape, ns in [((20, 6, 150), 1), ((66, 5, 31), 0), ((12, 9, 133), 1), ((128, 6, 40), 1)]:
        g = cm.grid_2nd(*shape, gc=(3, 3, 1), dtype=dtype)
        c = cm.Case(g, nscalars=ns, rho=rho)
        p = capi.MhhDiffParams(); p.cs = 0.23; p.tPr = 1./3.; p.surface_model = 1
        d = B.DevCase(be, c); f = d.fields()
        B.ok(be, be.lib.mhh_rhs_exec(d.G, adv, dif, C.byref(f), C.byref(p), be.stream))
        got = [be.host(x) for x in (d.ut, d.vt, d.wt)] + [be.host(x) for x in d.st]
        Gh = g.host_struct()
        ut, vt, wt = c.ut.copy(), c.vt.copy(), c.wt.copy(); st = [x.copy() for x in c.st]
        a = (ptr(c.u), ptr(c.v), ptr(c.w), ptr(c.rhoref), ptr(c.rhorefh))
        O.orc_advec_u(Gh, 25, ptr(ut), *a); O.orc_advec_v(Gh, 25, ptr(vt), *a); O.orc_advec_w(Gh, 25, ptr(wt), *a)
        for n in range(ns):
            O.orc_advec_s(Gh, 25, ptr(st[n]), ptr(c.s[n]), *a)
        v5 = dbl(1e-5)
        O.orc_smag2_diff_u(Gh, 1, ptr(ut), ptr(c.u), ptr(c.v), ptr(c.w), ptr(c.evisc), ptr(c.u_fluxbot), ptr(c.u_fluxtop), ptr(c.rhoref), ptr(c.rhorefh), v5)
        O.orc_smag2_diff_v(Gh, 1, ptr(vt), ptr(c.u), ptr(c.v), ptr(c.w), ptr(c.evisc), ptr(c.v_fluxbot), ptr(c.v_fluxtop), ptr(c.rhoref), ptr(c.rhorefh), v5)
        O.orc_smag2_diff_w(Gh, ptr(wt), ptr(c.u), ptr(c.v), ptr(c.w), ptr(c.evisc), ptr(c.rhoref), ptr(c.rhorefh), v5)
        for n in range(ns):
            O.orc_smag2_diff_c(Gh, 1, ptr(st[n]), ptr(c.s[n]), ptr(c.evisc), ptr(c.s_fluxbot), ptr(c.s_fluxtop), ptr(c.rhoref), ptr(c.rhorefh), dbl(1./3.), v5)
        for a_, b_, nm in zip(got, [ut, vt, wt] + st, ("ut", "vt", "wt", "st")):
            assert same(a_, b_), (shape, rho, nm, cm.ulp_diff(a_, b_))


@pytest.mark.parametrize("dtype", DTYPES)
def test_rhs44_marching_form_equals_cell_form(be, dtype):
    """k_march4.hip (LDS planes + register columns, the view-generic arithmetic of cell_ops.h) against the one-thread-per-cell
    Rhs44Op: same bits on tiles cut by the domain edge, several k-chunks, 2-D runs, with a scalar and folded buoyancy."""
    shapes = [(70, 10, 12), (16, 12, 70), (18, 1, 8), (17, 9, 8)] if dtype == np.float64 else [(74, 10, 12), (18, 9, 8), (16, 12, 10)]
    for shape in shapes:                      # (17, 9, 8) fp64 and (18, 9, 8) / (16, 12, 10) fp32: rows not 16-byte aligned -> 4-byte copies
        g = cm.grid_4th(*shape, dtype=dtype)
        c = cm.Case(g, nscalars=1)
        threfh = (300. + 0.37*np.arange(g.kcells)).astype(dtype)
        out = {}
        for impl in ("march", "cell"):
            d = B.DevCase(be, c); f = d.fields(); dth = be.arr(threfh)
            p = capi.MhhDiffParams(); p.buoyancy = 4; p.th_for_N2 = 0; p.threfh = be.ptr(dth).value; p.grav = 9.81
            os.environ["MHH_RHS44_IMPL"] = impl
            try:
                n0 = be.lib.mhh_stat_rhs44_march_launches()
                B.ok(be, be.lib.mhh_rhs_exec(d.G, cm.ADVEC_4, cm.DIFF_4, C.byref(f), C.byref(p), be.stream))
                ran = be.lib.mhh_stat_rhs44_march_launches() - n0
            finally:
                del os.environ["MHH_RHS44_IMPL"]
            out[impl] = [be.host(x) for x in (d.ut, d.vt, d.wt, d.st[0])]
            assert ran == (1 if impl == "march" else 0), (impl, shape, ran)
        for a, b, nm in zip(out["march"], out["cell"], ("ut", "vt", "wt", "st")):
            assert same(a, b), (shape, nm, cm.ulp_diff(a, b))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("sm,neutral", [(1, 0), (0, 0), (1, 1)])
def test_exec_viscosity_marching_form_equals_cell_form(be, sm, neutral, dtype):
    """k_visc.hip (LDS planes, carried vertical-shear terms) against the one-thread-per-cell ViscosityOp: the same bits,
    on tiles that are cut by the domain edge, several k-chunks, and layouts that do / do not allow LDS-DMA."""
    shapes = [(70, 10, 12), (16, 12, 131), (18, 9, 8), (17, 9, 8)] if dtype == np.float64 else [(70, 10, 12), (18, 9, 8), (16, 12, 10)]
    for shape in shapes:                      # (17, 9, 8) fp64, (70, ..) / (16, ..) fp32: not 16-byte aligned -> 4-byte copies
        g = cm.grid_2nd(*shape, gc=(3, 3, 1), dtype=dtype)
        c = cm.Case(g, periodic=True)
        thref = (300. + 7.*np.sin(0.37*np.arange(g.kcells))).astype(dtype)     # a different grav/thref[k] per level (one lane each, k_visc.hip)
        out = {}
        for impl in ("march", "cell"):
            d = B.DevCase(be, c); f = d.fields()
            p = capi.MhhDiffParams(); p.cs = 0.23; p.tPr = 1./3.; p.surface_model = sm; p.neutral = neutral
            p.N2 = None; p.th_for_N2 = 0; dthref = be.arr(thref); p.thref = be.ptr(dthref).value; p.grav = 9.81
            ml = B.mlen0(be, g, 0.23); p.mlen0 = be.ptr(ml).value
            os.environ["MHH_VISC_IMPL"] = impl
            try:
                n0 = be.lib.mhh_stat_visc_march_launches()
                B.ok(be, be.lib.mhh_diff_exec_viscosity(d.G, cm.DIFF_SMAG2, C.byref(f), C.byref(p), be.stream))
                ran = be.lib.mhh_stat_visc_march_launches() - n0
            finally:
                del os.environ["MHH_VISC_IMPL"]
            out[impl] = be.host(d.evisc)
            assert ran == (1 if impl == "march" else 0), (impl, shape, ran)     # 16-byte or 4-byte copies, never the cell form
        assert same(out["march"], out["cell"]), (shape, cm.ulp_diff(out["march"], out["cell"]))


@pytest.mark.gpu
def test_sqrt_in_range_equals_the_compilers_sqrt_on_the_device():
    """sqrt_in_range (csrc/gfx950_prims.h: the target's sqrt expansion without its range scaling and 0 / inf pass-through) against
    __builtin_sqrt on 2^28 arguments over [2^-767, 2^1023], one in sixteen within 1024 ulp of the 1e-9 the kernel's arguments
    are bounded by: not one differing bit."""
    be = B.get("hip")
    for seed in (1, 20240607):
        bad = C.c_ulonglong(12345)
        B.ok(be, be.lib.mhh_selftest_sqrt_in_range(1 << 28, seed, C.byref(bad), be.stream))
        assert bad.value == 0, (seed, bad.value)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("sm,neutral", [(1, 0), (0, 0), (1, 1)])
def test_exec_viscosity_with_the_mixing_length_table_equals_per_cell_evaluation(be, sm, neutral, dtype):
    """mhh_diff_params::mlen2 (mhh_smag2_mlen2_host: the squared mixing length per level for a horizontally uniform z0m,
    evaluated on the HOST) against the per-cell evaluation on the device: the same bits in both kernel forms -- IEEE
    division and square root round alike on either side."""
    for shape in [(70, 10, 12), (18, 9, 8)]:
        g = cm.grid_2nd(*shape, gc=(3, 3, 1), dtype=dtype)
        c = cm.Case(g, periodic=True)
        assert (c.z0m == c.z0m.flat[0]).all()
        thref = np.full(g.kcells, 300., dtype=dtype)
        out = {}
        for impl in ("march", "cell"):
            for table in (False, True):
                d = B.DevCase(be, c); f = d.fields()
                p = capi.MhhDiffParams(); p.cs = 0.23; p.tPr = 1./3.; p.surface_model = sm; p.neutral = neutral
                p.N2 = None; p.th_for_N2 = 0; dthref = be.arr(thref); p.thref = be.ptr(dthref).value; p.grav = 9.81
                ml_h = np.zeros(g.kcells, dtype=dtype)
                B.ok(be, be.lib.mhh_smag2_mlen0_host(g.host_struct(), 0.23, ptr(ml_h)))
                ml = be.arr(ml_h); p.mlen0 = be.ptr(ml).value
                if table:
                    m2_h = np.zeros(g.kcells, dtype=dtype)
                    B.ok(be, be.lib.mhh_smag2_mlen2_host(g.host_struct(), sm, neutral, ptr(ml_h), float(c.z0m.flat[0]), ptr(m2_h)))
                    assert (m2_h[g.kstart:g.kend] > 0).all()
                    m2 = be.arr(m2_h); p.mlen2 = be.ptr(m2).value
                os.environ["MHH_VISC_IMPL"] = impl
                try:
                    B.ok(be, be.lib.mhh_diff_exec_viscosity(d.G, cm.DIFF_SMAG2, C.byref(f), C.byref(p), be.stream))
                finally:
                    del os.environ["MHH_VISC_IMPL"]
                out[impl, table] = be.host(d.evisc)
        for impl in ("march", "cell"):
            assert same(out[impl, True], out[impl, False]), (impl, shape, cm.ulp_diff(out[impl, True], out[impl, False]))
        assert same(out["march", True], out["cell", True])


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("order", [2, 4])
def test_pres(be, order, dtype):
    O = cm.oracle()
    gl = [cm.grid_2nd(16, 12, 10, gc=(1, 1, 1), dtype=dtype), cm.grid_2nd(12, 10, 8, gc=(3, 3, 1), dtype=dtype), cm.grid_2nd(12, 1, 8, gc=(1, 1, 1), dtype=dtype),
          cm.grid_2nd(16, 8, 6, gc=(2, 2, 1), dtype=dtype)] if order == 2 \
        else [cm.grid_4th(16, 12, 12, dtype=dtype), cm.grid_4th(12, 1, 8, dtype=dtype), cm.grid_4th(16, 8, 8, dtype=dtype)]
    tol = 1e-11 if dtype == np.float64 else 2e-4
    for g in gl:
        c = cm.Case(g, rho=("random" if order == 2 else "one"), periodic=True)
        if order == 4:
            for m in (1, 2):
                c.w[g.kstart-m] = -c.w[g.kstart+m]; c.w[g.kend+m] = -c.w[g.kend-m]
        Gh = g.host_struct(); dt = 0.7
        d = B.DevCase(be, c); f = d.fields()
        plan = capi.PLAN()
        B.ok(be, be.lib.mhh_pres_plan_create(Gh, order, ptr(g.dz), ptr(g.dzhi), ptr(g.dzi4), ptr(g.dzhi4), ptr(c.rhoref), ptr(c.rhorefh), C.byref(plan)))
        try:
            # stage 1: input -- bit-exact incl. the ghost-cell side effects on ut, vt (and wt for pres_4)
            pk_want = np.zeros((g.ktot, g.jtot, g.itot), dtype=dtype)
            ut, vt, wt = c.ut.copy(), c.vt.copy(), c.wt.copy()
            O.orc_pres_input(Gh, order, ptr(pk_want), ptr(c.u), ptr(c.v), ptr(c.w), ptr(ut), ptr(vt), ptr(wt), ptr(c.rhoref), ptr(c.rhorefh), dbl(dt))
            pk = be.zeros((g.ktot, g.jtot, g.itot), dtype)
            B.ok(be, be.lib.mhh_pres_input(plan, d.G, C.byref(f), dt, be.ptr(pk), be.stream))
            assert same(be.host(pk), pk_want)
            assert same(be.host(d.ut), ut) and same(be.host(d.vt), vt) and same(be.host(d.wt), wt)
            # stage 2: solve
            p_want = np.zeros(g.shape3, dtype=dtype)
            pk_tmp = pk_want.copy()
            O.orc_pres_solve(Gh, order, ptr(p_want), ptr(pk_tmp), ptr(c.rhoref), ptr(c.rhorefh))
            B.ok(be, be.lib.mhh_pres_solve(plan, d.G, C.byref(f), be.ptr(pk), be.stream))
            p_got = be.host(d.p)
            kg = 1 if order == 2 else 2
            sl = (slice(g.kstart-kg, g.kend + (0 if order == 2 else 2)), slice(None) if g.jtot > 1 else slice(g.jstart, g.jend), slice(None))
            scale = np.abs(p_want).max()
            assert np.abs(p_got[sl] - p_want[sl]).max() <= tol*scale, np.abs(p_got[sl] - p_want[sl]).max()/scale
            # stage 3: output, from the oracle's p so that the stage itself is checked bit-exactly
            dp = be.arr(p_want); f2 = d.fields(); f2.p = be.ptr(dp).value
            O.orc_pres_output(Gh, order, ptr(ut), ptr(vt), ptr(wt), ptr(p_want))
            B.ok(be, be.lib.mhh_pres_output(plan, d.G, C.byref(f2), be.stream))
            assert same(be.host(d.ut), ut) and same(be.host(d.vt), vt) and same(be.host(d.wt), wt)
            # whole operator: projection leaves a divergence-free (u/dt + ut)
            d2 = B.DevCase(be, c); f3 = d2.fields()
            B.ok(be, be.lib.mhh_pres_exec(plan, d2.G, C.byref(f3), dt, be.stream))
            B.ok(be, be.lib.mhh_pres_input(plan, d2.G, C.byref(f3), dt, be.ptr(pk), be.stream))
            assert np.abs(be.host(pk)).max() <= (1e-9 if dtype == np.float64 else 2e-2) * np.abs(pk_want).max()
            # check_divergence reduction: bit-exact
            out = C.c_double(0)
            B.ok(be, be.lib.mhh_pres_check_divergence(d.G, order, C.byref(f), be.ptr(d.work), C.byref(out), be.stream))
            assert out.value == O.orc_pres_divergence(Gh, order, ptr(c.u), ptr(c.v), ptr(c.w), ptr(c.rhoref), ptr(c.rhorefh))
        finally:
            be.lib.mhh_pres_plan_destroy(plan)


@pytest.mark.parametrize("dtype", DTYPES)
def test_rk_substep_bitexact(be, dtype):
    O = cm.oracle()
    g = cm.grid_2nd(16, 12, 10, dtype=dtype)
    c = cm.Case(g)
    for order, nsub in ((3, 3), (4, 5)):
        for sub in range(nsub):
            a, at = c.u.copy(), c.ut.copy()
            O.orc_rk_substep(g.host_struct(), order, sub, dbl(0.31), ptr(a), ptr(at))
            da, dat = be.arr(c.u), be.arr(c.ut)
            B.ok(be, be.lib.mhh_rk_substep(be.grid(g), order, sub, 0.31, be.ptr(da), be.ptr(dat), be.stream))
            assert same(be.host(da), a) and same(be.host(dat), at), (order, sub)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("form", ["staged", "lds"])
def test_pres_exec_with_the_rk_substep_in_its_last_kernel(be, form, dtype):
    """mhh_pres_exec_rk = pres->exec(sub_dt) followed by timeloop.exec() for u, v, w (src/model.cxx:411,484) with the sub-step
    applied in the kernel that stores the corrected tendencies: the bits of mhh_pres_exec + mhh_rk_substep x 3 in both forms of
    Pres_2::exec, on every sub-step of RK3 and RK4 (the last one of a step and pres_4 take the separate kernels by themselves)."""
    cases = [(2, cm.grid_2nd(16, 8, 6, gc=(3, 3, 1), dtype=dtype), "random"), (2, cm.grid_2nd(32, 16, 9, gc=(1, 1, 1), dtype=dtype), "random")]
    if form == "staged":
        cases += [(2, cm.grid_2nd(12, 10, 8, gc=(3, 3, 1), dtype=dtype), "random"), (4, cm.grid_4th(16, 12, 12, dtype=dtype), "one")]
    for order, g, rho in cases:
        c = cm.Case(g, rho=rho, periodic=True); Gh = g.host_struct(); dt, sub_dt = 0.31, 0.1
        plan = capi.PLAN()
        B.ok(be, be.lib.mhh_pres_plan_create(Gh, order, ptr(g.dz), ptr(g.dzhi), ptr(g.dzi4), ptr(g.dzhi4), ptr(c.rhoref), ptr(c.rhorefh), C.byref(plan)))
        os.environ["MHH_PRES_LDS"] = "1" if form == "lds" else "0"
        try:
            for rko, nsub in ((3, 3), (4, 5)):
                for sub in range(nsub):
                    d1 = B.DevCase(be, c); f1 = d1.fields()
                    B.ok(be, be.lib.mhh_pres_exec(plan, d1.G, C.byref(f1), sub_dt, be.stream))
                    for a, at in ((d1.u, d1.ut), (d1.v, d1.vt), (d1.w, d1.wt)):
                        B.ok(be, be.lib.mhh_rk_substep(d1.G, rko, sub, dt, be.ptr(a), be.ptr(at), be.stream))
                    d2 = B.DevCase(be, c); f2 = d2.fields()
                    B.ok(be, be.lib.mhh_pres_exec_rk(plan, d2.G, C.byref(f2), sub_dt, rko, sub, dt, be.stream))
                    for nm in ("u", "v", "w", "ut", "vt", "wt", "p"):
                        assert same(be.host(getattr(d1, nm)), be.host(getattr(d2, nm))), (form, order, g.shape3, rko, sub, nm)
                    assert not same(be.host(d2.u), c.u)
        finally:
            os.environ.pop("MHH_PRES_LDS", None)
            be.lib.mhh_pres_plan_destroy(plan)


def test_errors_are_reported(be):
    g = cm.grid_2nd(16, 12, 10, gc=(1, 1, 1))
    c = cm.Case(g); d = B.DevCase(be, c)
    rc = be.lib.mhh_advec_u(d.G, cm.ADVEC_2I5, be.ptr(d.ut), be.ptr(d.u), be.ptr(d.v), be.ptr(d.w), be.ptr(d.rhoref), be.ptr(d.rhorefh), be.stream)
    assert rc == 1 and b"advec_2i5" in be.lib.mhh_last_error()
    with pytest.raises(capi.MhhError):
        B.ok(be, be.lib.mhh_advec_u(d.G, 7, be.ptr(d.ut), be.ptr(d.u), be.ptr(d.v), be.ptr(d.w), be.ptr(d.rhoref), be.ptr(d.rhorefh), be.stream))
    assert be.lib.mhh_diff_c(d.G, 2, None, be.ptr(d.u), 1e-5, be.stream) == 1


@pytest.mark.parametrize("dtype", DTYPES)
def test_slab_code_path_on_one_rank(be, dtype):
    """The slab kernels (N-S halo pack/unpack, pressure solve split at the transposes, 1-D rocFFT plans) run on one
    rank with the exchanges degenerated to local copies, against the regular single-rank path."""
    from microhh_amd.model import HotPath, synthetic_global
    shape = (16, 12, 10)
    dev = "cuda:0" if be.name == "hip" else "cpu"
    gi = synthetic_global("drycblles", *shape, dtype=dtype)
    a = HotPath("drycblles", *shape, dtype=dtype, device=dev, lib=be.lib, global_init=gi)
    b = HotPath("drycblles", *shape, dtype=dtype, device=dev, lib=be.lib, global_init=gi, force_slab=True)
    for hp in (a, b):
        hp.cyclic_prognostic(); hp.exec_viscosity(); hp.rhs()
    for n in ("u", "v", "w", "evisc", "ut", "vt", "wt"):
        assert same(be.host(getattr(a, n)), be.host(getattr(b, n))), n
    a.pres(); b.pres(); a.sync(); b.sync()
    tol = 1e-11 if dtype == np.float64 else 2e-4
    it = a.grid.interior
    for n in ("p", "ut", "vt", "wt"):
        x, y = be.host(getattr(a, n))[it], be.host(getattr(b, n))[it]
        assert np.abs(x - y).max() <= tol * np.abs(x).max(), (n, np.abs(x - y).max() / np.abs(x).max())
    a.close(); b.close()


def test_hotpath_restart_files_roundtrip(be, tmp_path):
    """HotPath.save / load: prognostic fields through the reference's field-file layout and back, bit for bit."""
    from microhh_amd.model import HotPath, synthetic_global
    shape = (16, 12, 10)
    dev = "cuda:0" if be.name == "hip" else "cpu"
    gi = synthetic_global("drycblles", *shape)
    a = HotPath("drycblles", *shape, device=dev, lib=be.lib, global_init=gi)
    b = HotPath("drycblles", *shape, device=dev, lib=be.lib, seed=99)
    a.save(str(tmp_path), 42)
    assert sorted(os.listdir(str(tmp_path))) == ["grid.0000000", "th.0000042", "u.0000042", "v.0000042", "w.0000042"]
    assert os.path.getsize(str(tmp_path / "u.0000042")) == 8 * 16 * 12 * 10
    b.load(str(tmp_path), 42)
    it = a.grid.interior
    for n in ("u", "v", "w"):
        assert np.array_equal(be.host(getattr(a, n))[it], be.host(getattr(b, n))[it]), n
    assert np.array_equal(be.host(a.s[0])[it], be.host(b.s[0])[it])
    for hp in (a, b):
        hp.cyclic_prognostic()
    for n in ("u", "v", "w"):                     # lateral ghost cells refilled from the loaded interiors
        assert np.array_equal(be.host(getattr(a, n))[it[0]], be.host(getattr(b, n))[it[0]]), n
    a.close(); b.close()
