"""Shared test plumbing: library loading, synthetic inputs, call helpers.

Oracle (oracle/liborc.so) and the in-place reference build (oracle/_ref/libmhhref.so) are TEST
infrastructure; they are loaded here and nowhere under microhh_amd/.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from microhh_amd.grid import (Grid, MhhGrid, EDGE_EW, EDGE_NS, EDGE_BOTH,  # noqa: E402,F401
                              ADVEC_2, ADVEC_2I5, ADVEC_2I4, ADVEC_2I62, ADVEC_2I53, ADVEC_4M, ADVEC_4, DIFF_2, DIFF_4, DIFF_SMAG2, moser_z, uniform_z)

ORACLE_DIR = os.path.join(ROOT, "oracle")
_libs = {}


def build_oracle():
    """make -C oracle (liborc*.so always; _ref only where /root/reference exists)."""
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "all"], check=True)


def _load(path):
    if path not in _libs:
        _libs[path] = C.CDLL(path)
    return _libs[path]


def oracle(perf=False):
    p = os.path.join(ORACLE_DIR, "liborc_perf.so" if perf else "liborc.so")
    if not os.path.exists(p):
        build_oracle()
    lib = _load(p)
    for n in ("orc_advec_cfl", "orc_smag2_dnmul", "orc_pres_divergence"):
        getattr(lib, n).restype = C.c_double
    return lib


def ref(perf=False):
    """The reference's own TUs compiled in place; None when neither source nor prebuilt lib exists."""
    p = os.path.join(ORACLE_DIR, "_ref", "libmhhref_perf.so" if perf else "libmhhref.so")
    if not os.path.exists(p):
        if os.path.isdir("/root/reference/src"):
            build_oracle()
        else:
            return None
    lib = _load(p)
    for n in ("ref_advec_2_cfl", "ref_advec_2i5_cfl", "ref_advec_2i4_cfl", "ref_advec_2i62_cfl", "ref_advec_2i53_cfl", "ref_advec_4m_cfl", "ref_advec_4_cfl", "ref_smag2_dnmul"):
        getattr(lib, n).restype = C.c_double
    return lib


def ptr(a):
    """void* of a numpy array (None -> NULL)."""
    if a is None:
        return C.c_void_p(0)
    assert a.flags["C_CONTIGUOUS"]
    return C.c_void_p(a.ctypes.data)


def dbl(x):
    return C.c_double(float(x))


class Case:
    """Synthetic inputs in the style of the reference's kernel_tuner harness (kernel_tuner/helpers.py:9,70-76:
    numpy.random.seed(666), uniform [0,1) fields in the order u, v, w, s, then rhoref, rhorefh), on a
    PHYSICAL grid instead of random metrics."""

    def __init__(self, grid, seed=666, nscalars=1, rho="random", tend_scale=1e-3, vel_shift=0.5, periodic=False):
        self.grid = g = grid
        t = g.np_dtype
        rs = np.random.RandomState(seed)
        n3, n2 = g.shape3, g.shape2

        def f3():
            return rs.random_sample(n3).astype(t)

        def f2(scale=1.0):
            return (rs.random_sample(n2) * scale).astype(t)
        # velocities centred around zero so that |u| upwind branches see both signs
        self.u = (f3() - t.type(vel_shift)).astype(t)
        self.v = (f3() - t.type(vel_shift)).astype(t)
        self.w = (f3() - t.type(vel_shift)).astype(t)
        self.s = [f3() for _ in range(nscalars)]
        if rho == "random":
            self.rhoref = (0.5 + rs.random_sample(g.kcells)).astype(t)
            self.rhorefh = (0.5 + rs.random_sample(g.kcells)).astype(t)
        else:
            self.rhoref = np.ones(g.kcells, dtype=t)
            self.rhorefh = np.ones(g.kcells, dtype=t)
        # walls: w = 0 at kstart and kend as the model keeps them (no-penetration)
        self.w[g.kstart] = 0
        self.w[g.kend] = 0
        self.ut = (f3() * tend_scale).astype(t)
        self.vt = (f3() * tend_scale).astype(t)
        self.wt = (f3() * tend_scale).astype(t)
        self.wt[g.kstart] = 0
        self.wt[g.kend:] = 0
        self.st = [(f3() * tend_scale).astype(t) for _ in range(nscalars)]
        self.evisc = (f3() * 0.1).astype(t)
        self.N2 = ((f3() - t.type(0.3)) * 1e-3).astype(t)
        self.dudz = f2(1e-2); self.dvdz = f2(1e-2); self.dbdz = f2(1e-4)
        self.z0m = np.full(n2, 0.1, dtype=t)
        self.u_fluxbot = f2(1e-2); self.u_fluxtop = f2(1e-2)
        self.v_fluxbot = f2(1e-2); self.v_fluxtop = f2(1e-2)
        self.s_fluxbot = f2(1e-2); self.s_fluxtop = f2(1e-2)
        self.p = np.zeros(n3, dtype=t)
        if periodic:   # what Boundary::set_prognostic_cyclic_bcs leaves behind (src/boundary.cxx:447-458)
            G = g.host_struct()
            for a in [self.u, self.v, self.w] + self.s:
                oracle().orc_boundary_cyclic(G, ptr(a), EDGE_BOTH)

    def copy_of(self, name):
        a = getattr(self, name)
        return [x.copy() for x in a] if isinstance(a, list) else a.copy()


def ulp_diff(a, b):
    """max |a-b| in units of the last place of max(|a|,|b|) (0 where bit-identical)."""
    a = np.asarray(a); b = np.asarray(b)
    d = np.abs(a.astype(np.float64) - b.astype(np.float64))
    m = np.maximum(np.abs(a), np.abs(b)).astype(a.dtype)
    sp = np.spacing(np.where(m == 0, np.finfo(a.dtype).tiny, m)).astype(np.float64)
    return float(np.max(d / sp)) if d.size else 0.0


SMALL_GRIDS_2 = [
    # (itot, jtot, ktot, igc, jgc, kgc)
    (16, 12, 10, 3, 3, 1),
    (24, 20, 18, 3, 3, 2),
    (8, 8, 6, 3, 3, 1),
]


def grid_2nd(itot=16, jtot=12, ktot=10, gc=(3, 3, 1), dtype=np.float64, stretched=True, **kw):
    z = None
    if stretched:
        z = moser_z(ktot, 1200.)
    return Grid(itot, jtot, ktot, 3200., 3200., 1200., order=2, igc=gc[0], jgc=gc[1], kgc=gc[2], z=z, dtype=dtype, **kw)


def grid_4th(itot=16, jtot=12, ktot=12, dtype=np.float64, **kw):
    return Grid(itot, jtot, ktot, 2*np.pi, np.pi, 2., order=4, z=moser_z(ktot, 2.), dtype=dtype, **kw)


def limiter_inputs(c, dtype):
    """Signed velocities (both upwind branches), a scalar with plateaus (the eps-guarded denominator), and with
    monotone as well as oscillating stretches (all pieces of the limiter function)."""
    u, v, w = (c.u - dtype(0.5)).astype(dtype), (c.v - dtype(0.5)).astype(dtype), (c.w - dtype(0.5)).astype(dtype)
    s = (np.round(c.s[0] * 6) / 6).astype(dtype)
    s.flat[::7] = c.s[0].flat[::7]
    return u, v, w, np.ascontiguousarray(s)
