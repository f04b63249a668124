"""Known-answer test of the assembled path (the reference's only one: cases/taylorgreen/taylorgreen_test.py:95-122):
2-D Taylor-Green vortex, free-slip walls, advec_2 + diff_2 + pres_2 + RK3 through the C ABI, compared with the
closed form  u = sin(2 pi x) cos(2 pi z) e^{-8 pi^2 nu t},  w = -cos(2 pi x) sin(2 pi z) e^{...},
p = (1/4 (cos 4 pi x + cos 4 pi z) - 1/4) e^{-16 pi^2 nu t}   (nu = (8 pi^2 1000)^-1, cases/taylorgreen/taylorgreen.ini:23-24).
The L1 error must be small and converge with second order between two grids. Also pins the vertical ghost-cell
kernels against the oracle."""
import ctypes as C

import numpy as np
import pytest

import backends as B
import common as cm
from common import ptr
from microhh_amd.model import HotPath, CASES, SURF

BACKENDS = [pytest.param("emul"), pytest.param("hip", marks=pytest.mark.gpu)]


@pytest.mark.parametrize("name", BACKENDS)
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_ghost_cells_bitexact(name, dtype):
    be = B.get(name); O = cm.oracle()
    for order, g in ((2, cm.grid_2nd(16, 12, 10, gc=(3, 3, 2), dtype=dtype)), (4, cm.grid_4th(16, 12, 12, dtype=dtype))):
        c = cm.Case(g)
        for bcb, bct in ((0, 0), (1, 1), (0, 1)):
            want = c.u.copy()
            O.orc_ghost_cells(g.host_struct(), order, ptr(want), bcb, bct, ptr(c.dudz), ptr(c.dvdz), ptr(c.u_fluxbot), ptr(c.u_fluxtop))
            d = be.arr(c.u)
            a = [be.arr(x) for x in (c.dudz, c.dvdz, c.u_fluxbot, c.u_fluxtop)]
            B.ok(be, be.lib.mhh_boundary_ghost_cells(be.grid(g), order, be.ptr(d), bcb, bct, *[be.ptr(x) for x in a], be.stream))
            assert np.array_equal(be.host(d), want) and not np.array_equal(want, c.u)
        if order == 4:
            for t in (0, 1):
                want = c.w.copy(); O.orc_ghost_cells_w(g.host_struct(), ptr(want), t)
                d = be.arr(c.w)
                B.ok(be, be.lib.mhh_boundary_ghost_cells_w(be.grid(g), be.ptr(d), t, be.stream))
                assert np.array_equal(be.host(d), want)


def run_tg(be, itot, ktot, T=0.2, dt=0.005, jtot=1, fused_rk=False):
    cfg = CASES["taylorgreen"]
    nu = cfg["visc"]
    dx, dz = 1./itot, 0.5/ktot
    x, xh = (np.arange(itot)+0.5)*dx, np.arange(itot)*dx
    z, zh = (np.arange(ktot)+0.5)*dz, np.arange(ktot)*dz
    gi = {"u": (np.sin(2*np.pi*xh)[None, None, :]*np.cos(2*np.pi*z)[:, None, None]) * np.ones((ktot, jtot, itot)),
          "w": (-np.cos(2*np.pi*x)[None, None, :]*np.sin(2*np.pi*zh)[:, None, None]) * np.ones((ktot, jtot, itot)),
          "v": np.zeros((ktot, jtot, itot))}
    for n in ("ut", "vt", "wt"):
        gi[n] = np.zeros((ktot, jtot, itot))
    for n in SURF:
        gi[n] = np.zeros((jtot, itot))
    dev = "cuda:0" if be.name == "hip" else "cpu"
    hp = HotPath("taylorgreen", itot, jtot, ktot, device=dev, lib=be.lib, global_init=gi, dt=dt)
    g = hp.grid
    zero2 = hp.surf["dudz"]                      # zero gradient at both walls (free slip)
    nsteps = int(round(T/dt))
    for _ in range(nsteps):
        for sub in range(3):
            hp.cyclic_prognostic()
            for f in (hp.u, hp.v):
                B.ok(be, be.lib.mhh_boundary_ghost_cells(hp.G, 2, f.data_ptr(), 1, 1, None, zero2.data_ptr(), None, zero2.data_ptr(), hp.stream))
            hp.rhs()
            # RK3 sub-step length (Timeloop::get_sub_time_step, src/timeloop.cxx:337-341): cB[sub]*dt
            sub_dt = (1./3., 15./16., 8./15.)[sub] * dt
            hp.dt = sub_dt
            if fused_rk:        # pres->exec + the sub-step of u, v, w in one call (mhh_pres_exec_rk): same bits as the four calls below
                hp.pres_rk(3, sub, dt)
                continue
            hp.pres()
            for a, at in ((hp.u, hp.ut), (hp.v, hp.vt), (hp.w, hp.wt)):
                B.ok(be, be.lib.mhh_rk_substep(hp.G, 3, sub, dt, a.data_ptr(), at.data_ptr(), hp.stream))
    hp.cyclic_prognostic()
    div = hp.divergence()
    hp.sync()
    it = g.interior
    u3 = (hp.u.cpu().numpy() if be.name == "hip" else hp.u.numpy())[it]
    w3 = (hp.w.cpu().numpy() if be.name == "hip" else hp.w.numpy())[it]
    p3 = (hp.p.cpu().numpy() if be.name == "hip" else hp.p.numpy())[it]
    v3 = (hp.v.cpu().numpy() if be.name == "hip" else hp.v.numpy())[it]
    hp.close()
    # the 3-D variant (jtot > 1, uniform in y, src/fields.cxx:995-996): every y-row carries the 2-D solution and v stays zero
    assert float(np.abs(v3).max()) < 1e-12
    for a3 in (u3, w3, p3):
        assert float(np.abs(a3 - a3[:, :1, :]).max()) < 1e-12
    u, w, p = u3[:, 0, :], w3[:, 0, :], p3[:, 0, :]
    dec = np.exp(-8*np.pi**2*nu*T)
    uref = np.sin(2*np.pi*xh)[None, :]*np.cos(2*np.pi*z)[:, None]*dec
    wref = -np.cos(2*np.pi*x)[None, :]*np.sin(2*np.pi*zh)[:, None]*dec
    pref = (0.25*(np.cos(4*np.pi*x)[None, :] + np.cos(4*np.pi*z)[:, None]) - 0.25)*dec**2
    p = p - p.mean() + pref.mean()               # pressure is defined up to a constant
    err = lambda a, b: float(np.sum(dx*dz*np.abs(a-b)))   # noqa: E731  (Get_error, taylorgreen_test.py:110-122)
    return err(u, uref), err(w, wref), err(p, pref), div


@pytest.mark.parametrize("name", BACKENDS)
def test_taylorgreen_known_answer_and_convergence(name):
    be = B.get(name)
    e1 = run_tg(be, 32, 16)
    e2 = run_tg(be, 64, 32)
    for n, a, b in zip("uwp", e1[:3], e2[:3]):
        order = np.log2(a/b)
        assert b < 2e-3 and 1.7 < order < 2.4, (n, a, b, order)
    assert e1[3] < 1e-10 and e2[3] < 1e-10        # the projected velocity is divergence free


@pytest.mark.parametrize("name", BACKENDS)
def test_taylorgreen_with_the_rk_substep_inside_the_pressure_kernel(name):
    """The same run with mhh_pres_exec_rk (Pres::exec + the Runge-Kutta sub-step of u, v, w in the kernel that stores the
    corrected tendencies): identical errors -- the calls are bit-identical -- on the 2-D grid and, on the GPU, at configs[0]'s 64^3."""
    be = B.get(name)
    assert run_tg(be, 32, 16, fused_rk=True) == run_tg(be, 32, 16)
    if name == "hip":
        assert run_tg(be, 64, 64, jtot=64, fused_rk=True) == run_tg(be, 64, 64, jtot=64)


@pytest.mark.gpu
def test_taylorgreen_3d_64_cubed_configs0():
    """BASELINE.json configs[0]: taylorgreen 64^3 (jtot = 64, uniform in y), advec_2 + diff_2 + pres_2 + RK3 on the GPU: the same
    closed form in every y-row, the error of the 64 x 32 two-dimensional run's class, divergence at rounding level."""
    be = B.get("hip")
    eu, ew, ep, div = run_tg(be, 64, 64, jtot=64)
    assert eu < 2e-3 and ew < 2e-3 and ep < 2e-3, (eu, ew, ep)
    assert div < 1e-10
    e2 = run_tg(be, 32, 32, jtot=32)
    for n, a, b in zip("uwp", e2[:3], (eu, ew, ep)):
        order = np.log2(a/b)
        assert 1.7 < order < 2.4, (n, a, b, order)
