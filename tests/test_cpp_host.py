"""The C++ host layer (microhh_amd/host/mhh_host.h: the reference's Advec/Diff/Pres/Boundary_cyclic interfaces over
the C ABI) driven from a C++ program the way Model::exec drives the reference operators, checked against the oracle."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

import common as cm
from common import ptr, dbl

CPP = os.path.join(cm.ROOT, "tests", "cpp")


def test_host_header_is_self_contained():
    """not gpu: the adaptor header compiles with a plain host compiler (no HIP, no torch types)."""
    src = '#include "%s/microhh_amd/host/mhh_host.h"\nint main() { mhh_host::Grid<double> g; (void)g; return 0; }\n' % cm.ROOT
    with tempfile.TemporaryDirectory() as tmp:
        p = os.path.join(tmp, "t.cpp")
        open(p, "w").write(src)
        subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", p], check=True)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4, 5, 8, 9, 25, 41, 57], ids=["two-calls", "fused", "two-calls-limited", "fused-limited", "two-calls-buoyancy", "fused-buoyancy",
                                                                          "slab-rccl-two-calls", "slab-rccl-fused", "slab-rccl-overlapped-halos", "slab-rccl-sliced-transposes",
                                                                          "slab-rccl-overlapped-and-sliced"])
def test_cpp_host_substep_matches_oracle(mode):
    """mode bit 0: advec->exec + diff->exec as the fused pass; bit 1: "th" in advec.fluxlimit_list (kgc = 2);
    bit 2: Thermo_dry buoyancy (thermo->exec before advec, or folded into the fused pass); bit 3: the slab code path of
    microhh_amd/host/mhh_host_rccl.h (north-south halos by ncclSend / ncclRecv, the pressure solve around two grouped all-to-alls,
    maxima by ncclAllReduce) on a one-rank RCCL communicator -- same oracle, same tolerances; bit 4: the overlapped sub-step of the
    C++ driver (Substep_slab: prognostic halos on an exchange stream while the interior rows are worked, both edge strips in one
    launch per operator); bit 5: the pressure solve in four k-slices whose all-to-alls run on a second stream (Pres_slab::set_chunks).
    The grid has 24 rows = three strips of eight, so the slab solves take the x stages with the transforms in LDS."""
    subprocess.run(["make", "-s", "-C", CPP], check=True)
    g = cm.grid_2nd(32, 24, 16, gc=(3, 3, 2 if mode & 2 else 1), stretched=False)
    c = cm.Case(g, rho="one")
    sm, dt, visc = 1, 0.6, 1e-5
    thref = np.full(g.kcells, 300.)
    threfh = 300. + 0.37*np.arange(g.kcells)
    with tempfile.TemporaryDirectory() as tmp:
        fin, fout = os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin")
        with open(fin, "wb") as f:
            np.array([g.itot, g.jtot, g.ktot, g.igc, g.jgc, g.kgc, sm, mode], dtype=np.int32).tofile(f)
            np.array([g.xsize, g.ysize, g.zsize, dt, visc, visc], dtype=np.float64).tofile(f)
            for a in (g.z, g.zh, g.dz, g.dzh, g.dzi, g.dzhi, g.dzi4, g.dzhi4, c.rhoref, c.rhorefh, c.u, c.v, c.w, c.s[0], c.ut, c.vt, c.wt, c.st[0],
                      c.u_fluxbot, c.u_fluxtop, c.v_fluxbot, c.v_fluxtop, c.s_fluxbot, c.s_fluxtop, c.dudz, c.dvdz, c.dbdz, c.z0m, thref) + ((threfh,) if mode & 4 else ()):
                np.ascontiguousarray(a, dtype=np.float64).tofile(f)
        r = subprocess.run([os.path.join(CPP, "host_step"), fin, fout], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        raw = np.fromfile(fout, dtype=np.float64)
    cfl, dnum, div = raw[:3]
    n3 = g.ncells
    got = {n: raw[3 + m*n3: 3 + (m+1)*n3].reshape(g.shape3) for m, n in enumerate(("ut", "vt", "wt", "tht", "evisc", "p"))}
    # oracle, same call sequence
    O = cm.oracle(); G = g.host_struct()
    for a in (c.u, c.v, c.w, c.s[0]):
        O.orc_boundary_cyclic(G, ptr(a), cm.EDGE_BOTH)
    ev = np.zeros(g.shape3); n2 = np.zeros(g.shape3)
    O.orc_smag2_strain2(G, sm, ptr(ev), ptr(c.u), ptr(c.v), ptr(c.w), ptr(c.dudz), ptr(c.dvdz))
    O.orc_calc_N2(G, ptr(n2), ptr(c.s[0]), ptr(thref), dbl(9.81))
    O.orc_smag2_evisc(G, sm, ptr(ev), ptr(n2), ptr(c.dbdz), ptr(c.z0m), dbl(0.23), dbl(1./3.))
    assert cfl == O.orc_advec_cfl(G, cm.ADVEC_2I5, ptr(c.u), ptr(c.v), ptr(c.w), dbl(dt))
    it = g.interior
    assert cm.ulp_diff(got["evisc"][it], ev[it]) <= 8
    ut, vt, wt, tht = c.ut.copy(), c.vt.copy(), c.wt.copy(), c.st[0].copy()
    evg = got["evisc"]          # continue from the device's evisc so that the stencil stages can be compared bit for bit
    a = (ptr(c.u), ptr(c.v), ptr(c.w), ptr(c.rhoref), ptr(c.rhorefh))
    if mode & 4:
        O.orc_buoyancy_tend(G, 2, ptr(wt), ptr(c.s[0]), ptr(threfh), dbl(9.81))
    O.orc_advec_u(G, 25, ptr(ut), *a); O.orc_advec_v(G, 25, ptr(vt), *a); O.orc_advec_w(G, 25, ptr(wt), *a)
    if mode & 2:
        O.orc_advec_s_lim(G, ptr(tht), ptr(c.s[0]), *a)
    else:
        O.orc_advec_s(G, 25, ptr(tht), ptr(c.s[0]), *a)
    O.orc_smag2_diff_u(G, sm, ptr(ut), ptr(c.u), ptr(c.v), ptr(c.w), ptr(evg), ptr(c.u_fluxbot), ptr(c.u_fluxtop), ptr(c.rhoref), ptr(c.rhorefh), dbl(visc))
    O.orc_smag2_diff_v(G, sm, ptr(vt), ptr(c.u), ptr(c.v), ptr(c.w), ptr(evg), ptr(c.v_fluxbot), ptr(c.v_fluxtop), ptr(c.rhoref), ptr(c.rhorefh), dbl(visc))
    O.orc_smag2_diff_w(G, ptr(wt), ptr(c.u), ptr(c.v), ptr(c.w), ptr(evg), ptr(c.rhoref), ptr(c.rhorefh), dbl(visc))
    O.orc_smag2_diff_c(G, sm, ptr(tht), ptr(c.s[0]), ptr(evg), ptr(c.s_fluxbot), ptr(c.s_fluxtop), ptr(c.rhoref), ptr(c.rhorefh), dbl(1./3.), dbl(visc))
    assert np.array_equal(got["tht"], tht)                      # scalar tendency untouched by the pressure step: bit-exact
    p = np.zeros(g.shape3); pk = np.zeros((g.ktot, g.jtot, g.itot))
    O.orc_pres_exec(G, 2, ptr(p), ptr(pk), ptr(c.u), ptr(c.v), ptr(c.w), ptr(ut), ptr(vt), ptr(wt), ptr(c.rhoref), ptr(c.rhorefh), dbl(dt))
    for n, w_ in (("ut", ut), ("vt", vt), ("wt", wt), ("p", p)):
        assert np.abs(got[n][it] - w_[it]).max() <= 1e-10*np.abs(w_[it]).max(), n
    assert div == O.orc_pres_divergence(G, 2, ptr(c.u), ptr(c.v), ptr(c.w), ptr(c.rhoref), ptr(c.rhorefh))
    assert dnum > 0
