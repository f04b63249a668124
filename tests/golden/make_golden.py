#!/usr/bin/env python3
"""Generate tests/golden/ref_vectors.npz: outputs of the REFERENCE's own kernels (oracle/_ref = the reference
translation units compiled in place from /root/reference, flags -O2 -ffp-contract=off) on seeded synthetic inputs.

Only data is stored (expected output arrays, interior cells); the inputs are regenerated from the seed by
tests/common.Case, which uses numpy.random.RandomState (bit-stable across numpy versions).
Run here (the reference cannot travel to the GPU box):  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import common as cm  # noqa: E402
from common import ptr, dbl  # noqa: E402

SEED = 666
GRID2 = (12, 10, 8, 3, 3, 1)
GRID4 = (12, 10, 8)


def cases(z2=None, z4=None):
    """The two input cases. The stretched grids come from tanh / log (moser_z), whose last bits depend on the host's numpy SIMD
    dispatch: the fixture stores the two z profiles it was made with (z2, z4) and the tests rebuild the grids from those,
    so that the vectors do not depend on the machine the tests run on. Everything else is RNG + exact arithmetic."""
    from microhh_amd.grid import Grid
    g2 = cm.grid_2nd(*GRID2[:3], gc=GRID2[3:])
    g4 = cm.grid_4th(*GRID4)
    if z2 is not None:
        g2 = Grid(GRID2[0], GRID2[1], GRID2[2], g2.xsize, g2.ysize, g2.zsize, order=2, igc=GRID2[3], jgc=GRID2[4], kgc=GRID2[5], z=np.asarray(z2, dtype=np.float64))
    if z4 is not None:
        g4 = Grid(GRID4[0], GRID4[1], GRID4[2], g4.xsize, g4.ysize, g4.zsize, order=4, z=np.asarray(z4, dtype=np.float64))
    return g2, cm.Case(g2, seed=SEED), g4, cm.Case(g4, seed=SEED)


def main():
    R = cm.ref()
    assert R is not None, "needs /root/reference (oracle/_ref)"
    g2, c2, g4, c4 = cases()
    out = {}
    for scheme, name, g, c in ((2, "ref_advec_2", g2, c2), (25, "ref_advec_2i5", g2, c2), (24, "ref_advec_2i4", g2, c2), (262, "ref_advec_2i62", g2, c2), (253, "ref_advec_2i53", g2, c2), (4, "ref_advec_4", g4, c4), (41, "ref_advec_4m", g4, c4)):
        G = g.host_struct()
        for comp, tn in enumerate(("ut", "vt", "wt")):
            t = c.copy_of(tn)
            getattr(R, name)(G, comp, ptr(t), None, ptr(c.u), ptr(c.v), ptr(c.w), ptr(c.rhoref), ptr(c.rhorefh))
            out["advec%d_%s" % (scheme, tn)] = t[g.interior]
        t = c.st[0].copy()
        getattr(R, name)(G, 3, ptr(t), ptr(c.s[0]), ptr(c.u), ptr(c.v), ptr(c.w), ptr(c.rhoref), ptr(c.rhorefh))
        out["advec%d_st" % scheme] = t[g.interior]
        out["advec%d_cfl" % scheme] = np.array(getattr(R, name + "_cfl")(G, ptr(c.u), ptr(c.v), ptr(c.w), dbl(0.37)))
    for order, g, c in ((2, g2, c2), (4, g4, c4)):
        G = g.host_struct()
        for is_w, src, tn in ((0, c.u, "ut"), (1, c.w, "wt")):
            t = c.copy_of(tn)
            getattr(R, "ref_diff_%d" % order)(G, is_w, ptr(t), ptr(src), dbl(1.3e-2))
            out["diff%d_%s" % (order, tn)] = t[g.interior]
    G = g2.host_struct(); c = c2
    for sm in (0, 1):
        s2 = np.zeros(g2.shape3)
        R.ref_smag2_strain2(G, sm, ptr(s2), ptr(c.u), ptr(c.v), ptr(c.w), ptr(c.dudz), ptr(c.dvdz))
        out["smag%d_strain2" % sm] = s2[g2.interior]
        for comp, tn, fb, ft in ((0, "ut", c.u_fluxbot, c.u_fluxtop), (1, "vt", c.v_fluxbot, c.v_fluxtop), (2, "wt", None, None)):
            t = c.copy_of(tn)
            R.ref_smag2_diff_uvw(G, comp, sm, ptr(t), ptr(c.u), ptr(c.v), ptr(c.w), ptr(c.evisc), ptr(fb), ptr(ft), ptr(c.rhoref), ptr(c.rhorefh), dbl(1e-5))
            out["smag%d_%s" % (sm, tn)] = t[g2.interior]
        t = c.st[0].copy()
        R.ref_smag2_diff_c(G, sm, ptr(t), ptr(c.s[0]), ptr(c.evisc), ptr(c.s_fluxbot), ptr(c.s_fluxtop), ptr(c.rhoref), ptr(c.rhorefh), dbl(1./3.), dbl(1e-5))
        out["smag%d_st" % sm] = t[g2.interior]
    # Koren-limited scalar advection (include/advec_monotonic.h:79-180): signed velocities, scalar with plateaus
    ul, vl, wl, sl = cm.limiter_inputs(c, np.float64)
    t = c.st[0].copy()
    R.ref_advec_s_lim(G, ptr(t), ptr(sl), ptr(ul), ptr(vl), ptr(wl), ptr(c.rhoref), ptr(c.rhorefh))
    out["advec_s_lim_st"] = t[g2.interior]
    out["smag_dnmul"] = np.array(R.ref_smag2_dnmul(G, ptr(c.evisc), dbl(1./3.)))
    # input fingerprint so that a drift of the input recipe is detected rather than misread as a kernel error
    out["fingerprint"] = np.array([c2.u.sum(), c2.rhorefh.sum(), c4.w.sum(), c2.evisc.sum()])
    out["z2"] = g2.z[g2.kstart:g2.kend].astype(np.float64); out["z4"] = g4.z[g4.kstart:g4.kend].astype(np.float64)
    np.savez_compressed(os.path.join(HERE, "ref_vectors.npz"), **out)
    print("wrote %d arrays" % len(out))


if __name__ == "__main__":
    main()
