"""BASELINE.json's full sizes on the GPU, through size-independent properties (the oracle finishes only the small cases):
two independent kernel forms agree bit for bit (k-marching LDS kernels vs one-thread-per-cell kernels, fused vs
per-operator launches), the pressure step is a projection (divergence after it is rounding-level and a second solve
finds nothing left), the cyclic fill is idempotent, the slab code path reproduces the single-rank bits."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _hp(case, shape, **kw):
    from microhh_amd.model import HotPath
    if case == "gabls1":                       # BASELINE.json configs[4] is single precision
        kw.setdefault("dtype", np.float32)
    return HotPath(case, *shape, device="cuda:0", **kw)


def _tend(hp):
    return [hp.ut, hp.vt, hp.wt] + list(hp.st)


def _run_rhs(hp, fn, env=None):
    import torch
    keep = [t.clone() for t in _tend(hp)]
    for k, v in (env or {}).items():
        os.environ[k] = v
    try:
        fn()
        hp.sync()
    finally:
        for k in (env or {}):
            os.environ.pop(k, None)
    out = [t.clone() for t in _tend(hp)]
    for t, k in zip(_tend(hp), keep):
        t.copy_(k)
    del keep
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("case,shape", [("drycblles", (256, 256, 256)), ("drycblles", (512, 512, 512)), ("moser600", (512, 256, 256)),
                                        ("gabls1", (1024, 128, 256)), ("gabls1", (1024, 1024, 256))],
                         ids=["configs1-drycblles256", "configs3-drycblles512", "configs2-moser600", "configs4-gabls1-one-rank-of-8-fp32", "configs4-gabls1-1024x1024x256-fp32"])
def test_kernel_forms_agree_at_full_size(case, shape):
    import torch
    hp = _hp(case, shape)
    hp.cyclic_prognostic()
    # exec_viscosity: marching form vs cell form
    if case in ("drycblles", "gabls1"):
        hp.exec_viscosity(); hp.sync(); ev_march = hp.evisc.clone()
        os.environ["MHH_VISC_IMPL"] = "cell"
        try:
            hp.exec_viscosity(); hp.sync()
        finally:
            del os.environ["MHH_VISC_IMPL"]
        assert torch.equal(ev_march, hp.evisc)
        assert float(hp.evisc[hp.grid.kstart:hp.grid.kend].min()) > 0.
        del ev_march
    fused = _run_rhs(hp, hp.rhs)                                   # k-marching kernels
    unfused = _run_rhs(hp, hp.rhs_unfused)                         # Advec::exec + Diff::exec: the marching kernels, one operator each
    names = ["ut", "vt", "wt", "st"]
    for a, b, n in zip(fused, unfused, names):
        assert torch.equal(a, b), (case, shape, n, float((a - b).abs().max()))
    perfield = _run_rhs(hp, hp.rhs_unfused, env={"MHH_ADVEC25_IMPL": "cell", "MHH_DIFF22_IMPL": "cell", "MHH_RHS44_IMPL": "cell"})   # one cell kernel per field
    for a, b, n in zip(fused, perfield, names):
        assert torch.equal(a, b), (case, shape, n, "per-field kernels")
    del perfield
    cell = _run_rhs(hp, hp.rhs, env={"MHH_RHS25_IMPL": "cell", "MHH_RHS44_IMPL": "cell"})   # fused cell kernels
    for a, b, n in zip(fused, cell, names):
        assert torch.equal(a, b), (case, shape, n, "fused cell form")
    assert not torch.equal(fused[0], hp.ut)                       # the pass did something
    hp.close()


@pytest.mark.parametrize("case,shape", [("drycblles", (256, 256, 256)), ("drycblles", (512, 512, 512)), ("moser600", (512, 256, 256))],
                         ids=["pres_2-256", "pres_2-512-configs3", "pres_4"])
def test_pressure_step_is_a_projection_at_full_size(case, shape):
    import torch
    hp = _hp(case, shape, dt=0.5)
    g = hp.grid
    if case == "moser600":          # conservation-type w ghost cells, as Boundary::set_ghost_cells_w sets them before pres->exec
        for m in (1, 2):
            hp.w[g.kstart-m] = -hp.w[g.kstart+m]; hp.w[g.kend+m] = -hp.w[g.kend-m]
    hp.cyclic_prognostic()
    if case == "drycblles":
        hp.exec_viscosity()
    hp.rhs()
    it = (slice(g.kstart, g.kend), slice(g.jstart, g.jend), slice(g.istart, g.iend))
    scale = max(float(t[it].abs().max()) for t in (hp.ut, hp.vt, hp.wt)) + max(float(t[it].abs().max()) for t in (hp.u, hp.v, hp.w)) / hp.dt
    hp.pres(); hp.sync()
    p1 = hp.p.clone()
    # second solve on the projected tendencies: nothing left to remove
    hp.pres(); hp.sync()
    dmin = min(float(g.dx), float(g.dy), float(np.min(g.dz[g.kstart:g.kend])))
    assert float(hp.p[it].abs().max()) <= 1e-9 * max(1.0, float(p1[it].abs().max())), (float(hp.p[it].abs().max()), float(p1[it].abs().max()))
    assert float(p1[it].abs().max()) > 0 and np.isfinite(scale) and dmin > 0
    hp.close()


@pytest.mark.parametrize("case,shape,tol", [("drycblles", (512, 512, 512), 1e-11), ("gabls1", (1024, 1024, 256), 2e-4), ("moser600", (512, 256, 256), 1e-11)],
                         ids=["configs3-drycblles512-fp64", "configs4-gabls1-fp32", "configs2-moser600-fp64-pres_4"])
def test_pressure_lds_transform_form_matches_staged_form_at_full_size(case, shape, tol):
    """Pres_2::exec / Pres_4::exec with the transforms in LDS (three kernels, the form mhh_pres_exec takes by itself at these sizes)
    against the staged rocFFT form on the same right-hand side: p and the corrected tendencies within the pressure tolerance."""
    import torch
    hp = _hp(case, shape, dt=0.5)
    assert hp.lib.mhh_pres_plan_has_lds_form(hp.plan) == 1
    hp.cyclic_prognostic(); hp.exec_viscosity(); hp.rhs(); hp.sync()
    keep = [t.clone() for t in (hp.ut, hp.vt, hp.wt)]
    out = {}
    for form in ("staged", "default"):
        for t, k in zip((hp.ut, hp.vt, hp.wt), keep):
            t.copy_(k)
        hp.p.zero_()
        if form == "staged":
            os.environ["MHH_PRES_LDS"] = "0"
        try:
            hp.pres(); hp.sync()
        finally:
            os.environ.pop("MHH_PRES_LDS", None)
        out[form] = [t.clone() for t in (hp.p, hp.ut, hp.vt, hp.wt)]
    del keep
    assert not torch.equal(out["staged"][0], out["default"][0])          # two different sets of transforms
    for a, b, n in zip(out["staged"], out["default"], ("p", "ut", "vt", "wt")):
        assert float((a - b).abs().max()) <= tol * float(a.abs().max()), (n, float((a - b).abs().max()) / float(a.abs().max()))
    hp.close()


def test_cyclic_fill_is_idempotent_and_periodic_at_full_size():
    import torch
    hp = _hp("drycblles", (256, 256, 256))
    g = hp.grid
    hp.cyclic_prognostic(); hp.sync()
    once = [t.clone() for t in (hp.u, hp.s[0])]
    hp.cyclic_prognostic(); hp.sync()
    for a, b in zip(once, (hp.u, hp.s[0])):
        assert torch.equal(a, b)
    u = hp.u
    assert torch.equal(u[:, :, :g.igc], u[:, :, g.iend-g.igc:g.iend]) and torch.equal(u[:, :, g.iend:], u[:, :, g.istart:g.istart+g.igc])
    assert torch.equal(u[:, :g.jgc, :], u[:, g.jend-g.jgc:g.jend, :]) and torch.equal(u[:, g.jend:, :], u[:, g.jstart:g.jstart+g.jgc, :])
    hp.close()


def test_slab_code_path_matches_single_rank_bits_at_256():
    """One rank, slab kernels (halo pack/unpack, evisc on ghost rows, split pressure solve) against the plain path."""
    import torch
    from microhh_amd.model import synthetic_global
    shape = (256, 64, 128)
    gi = synthetic_global("drycblles", *shape)
    a = _hp("drycblles", shape, global_init=gi)
    b = _hp("drycblles", shape, global_init=gi, force_slab=True)
    for hp in (a, b):
        hp.cyclic_prognostic(); hp.exec_viscosity(); hp.rhs(); hp.sync()
    g = a.grid
    it = (slice(g.kstart, g.kend), slice(g.jstart, g.jend), slice(g.istart, g.iend))
    for n in ("evisc", "ut", "vt", "wt"):
        assert torch.equal(getattr(a, n)[it], getattr(b, n)[it]), n
    # the overlapped order: exchange on its own stream, interior rows meanwhile, edge rows after (fresh fields each time)
    for _ in range(3):
        c = _hp("drycblles", shape, global_init=gi, force_slab=True, overlap=True)
        assert c.can_overlap
        c.halo_visc_rhs(); c.sync()
        for n in ("evisc", "ut", "vt", "wt"):
            assert torch.equal(getattr(a, n)[it], getattr(c, n)[it]), ("overlapped", n)
        assert torch.equal(a.st[0][it], c.st[0][it])
        c.close()
    a.pres(); b.pres(); a.sync(); b.sync()
    for n in ("p", "ut", "vt", "wt"):
        x, y = getattr(a, n)[it], getattr(b, n)[it]
        assert float((x - y).abs().max()) <= 1e-11 * float(x.abs().max()), n
    a.close(); b.close()


@pytest.mark.parametrize("case,shape", [("drycblles", (256, 256, 256)), ("moser600", (256, 128, 128)), ("gabls1", (1024, 128, 256)), ("drycblles", (512, 512, 512))],
                         ids=["2i5-smag2-pres_2", "4-4-pres_4", "gabls1-fp32-one-rank-of-8", "configs3-512-pres_2-with-transforms-in-LDS"])
def test_substep_is_deterministic(case, shape):
    """The marching kernels order their LDS-DMA copies, deferred stores and prefetched tendencies themselves (inline asm, no
    compiler-placed waits): the same sub-step from the same inputs must give the same bits every time."""
    import torch
    hp = _hp(case, shape)
    state = [hp.ut, hp.vt, hp.wt, hp.p, hp.evisc] + list(hp.st)
    init = [t.clone() for t in state]

    def run():
        for t, k in zip(state, init):
            t.copy_(k)
        hp.step()
    run(); hp.sync()
    ref = [t.clone() for t in state]
    for n in range(20 if (case == "gabls1" or shape[0] == 512) else 60):
        run()
        assert all(torch.equal(a, b) for a, b in zip(state, ref)), n
    hp.close()


@pytest.mark.parametrize("case,shape", [("drycblles", (64, 64, 64)), ("drycblles", (256, 256, 256)), ("moser600", (256, 128, 128))],
                         ids=["64-cubed", "configs1-drycblles256", "pres_4"])
def test_step_replayed_as_hip_graph_gives_the_same_bits(case, shape):
    """A step captured into a hipGraph (HotPath.capture_step: the library only enqueues on the caller's stream) and replayed
    from the same inputs equals the step launched call by call, bit for bit -- also after the fields changed in between."""
    import torch
    hp = _hp(case, shape)
    state = [hp.ut, hp.vt, hp.wt, hp.p, hp.evisc] + list(hp.st)
    init = [t.clone() for t in state]

    def reset(scale=1.0):
        for t, k in zip(state, init):
            t.copy_(k * scale)
    reset(); hp.step(); hp.sync()
    eager = [t.clone() for t in state]
    graph = hp.capture_step()
    for rep in range(3):
        reset(); graph.replay(); hp.sync()
        assert all(torch.equal(a, b) for a, b in zip(state, eager)), rep
    reset(0.5); hp.step(); hp.sync()
    eager2 = [t.clone() for t in state]
    reset(0.5); graph.replay(); hp.sync()
    assert all(torch.equal(a, b) for a, b in zip(state, eager2))
    assert not torch.equal(eager[0], eager2[0])
    hp.close()
