"""The second, NAMED build (microhh_amd/libmhh_hip_fma.so: marching kernels compiled with FMA contraction allowed) against the
bit-exact default build and the CPU oracle. It is not bit-identical by construction; its stated tolerance is

    |increment_fma - increment_exact|  <=  TOL_ULP ulp of the field's largest |increment|

per tendency (increment = tendency after the fused pass - tendency before), fp64. The default library stays the product's
reference for parity; this build exists for the throughput line bench.py reports next to it (`fma_build`)."""
import ctypes as C
import os

import numpy as np
import pytest

import backends as B
import common as cm
from microhh_amd import capi

TOL_ULP = 64.0
FMA = os.path.join(cm.ROOT, "microhh_amd", "libmhh_hip_fma.so")


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(70, 10, 40), (128, 64, 140)])
def test_fma_build_within_stated_tolerance(shape):
    be = B.get("hip")
    assert os.path.exists(FMA), "the FMA build is part of __graft_entry__.build()"
    fl = capi.bind(C.CDLL(FMA))
    g = cm.grid_2nd(*shape, gc=(3, 3, 1))
    c = cm.Case(g, nscalars=1, rho="one")
    p = capi.MhhDiffParams(); p.cs = 0.23; p.tPr = 1./3.; p.surface_model = 1
    out = {}
    for name, lib in (("exact", be.lib), ("fma", fl)):
        d = B.DevCase(be, c); f = d.fields()
        rc = lib.mhh_rhs_exec(d.G, cm.ADVEC_2I5, cm.DIFF_SMAG2, C.byref(f), C.byref(p), be.stream)
        assert rc == 0, lib.mhh_last_error()
        out[name] = [be.host(x) for x in (d.ut, d.vt, d.wt, d.st[0])]
    before = [c.ut, c.vt, c.wt, c.st[0]]
    it = g.interior
    worst = 0.0
    for a, b, t0, nm in zip(out["exact"], out["fma"], before, ("ut", "vt", "wt", "st")):
        inc = (a - t0)[it]
        scale = np.abs(inc).max()
        err = np.abs((b - a)[it]).max() / np.spacing(scale)
        worst = max(worst, err)
        assert err <= TOL_ULP, (shape, nm, err)
    # the named build must BE a different build: on a grid of this size contraction changes the last bits of some tendency (if the
    # link step had silently produced the bit-exact objects the tolerance check above would pass trivially)
    if shape[0]*shape[1]*shape[2] >= 128*64*140:
        assert any(not np.array_equal(a, b) for a, b in zip(out["exact"], out["fma"])), "the FMA library gives the bit-exact build's bits"
    print("fma build: worst deviation %.1f ulp of the largest increment (stated tolerance %g)" % (worst, TOL_ULP))
