"""Host-side driver of one RK sub-step's hot path: the slice of ``Model<TF>::exec`` (src/model.cxx:346-411)
that this package accelerates, expressed as calls into the C ABI.

    boundary->set_prognostic_cyclic_bcs   -> mhh_boundary_cyclic_n          (src/model.cxx:346)
    diff->exec_viscosity(thermo)          -> mhh_diff_exec_viscosity         (:354)
    advec->exec ; diff->exec              -> mhh_rhs_exec (fused, same bits) (:388, :392)
    pres->exec(dt)                        -> mhh_pres_exec                   (:411)

PyTorch is used for device memory and streams only; every kernel is the hand-written HIP in csrc/.
Case recipes follow SURVEY.md §8(d): synthetic drycblles / taylorgreen / moser600 shaped inputs.
"""
import ctypes as C
import math

import numpy as np

from . import capi
from .grid import Grid, ADVEC_2, ADVEC_2I5, ADVEC_4, DIFF_2, DIFF_4, DIFF_SMAG2, EDGE_BOTH, moser_z

CASES = {
    # name: (advec, diff, pres order, spatial order, ghost cells, domain, scalars, surface model)
    "taylorgreen": dict(advec=ADVEC_2, diff=DIFF_2, pres=2, order=2, gc=(1, 1, 1), size=(1., 1., 0.5), nscalars=0, sm=0, visc=(8.*math.pi**2*1000.)**-1),
    "drycblles": dict(advec=ADVEC_2I5, diff=DIFF_SMAG2, pres=2, order=2, gc=(3, 3, 1), size=(3200., 3200., 1200.), nscalars=1, sm=1, visc=1e-5),
    "moser600": dict(advec=ADVEC_4, diff=DIFF_4, pres=4, order=4, gc=(3, 3, 3), size=(2*math.pi, math.pi, 2.), nscalars=0, sm=0, visc=1e-5),
}


def _t(torch, dtype):
    return torch.float64 if np.dtype(dtype) == np.float64 else torch.float32


class HotPath:
    """Device-resident fields of one rank + the operator calls of one sub-step (single GPU)."""

    def __init__(self, case, itot, jtot, ktot, dtype=np.float64, device="cuda:0", seed=666, dt=1.0):
        import torch
        self.torch = torch
        self.lib = capi.lib()
        self.cfg = cfg = CASES[case]
        self.case = case
        self.dt = dt
        self.device = torch.device(device)
        z = moser_z(ktot, cfg["size"][2]) if case == "moser600" else None
        self.grid = g = Grid(itot, jtot, ktot, *cfg["size"], order=cfg["order"], igc=cfg["gc"][0], jgc=cfg["gc"][1], kgc=cfg["gc"][2], z=z, dtype=dtype)
        self.G = g.device_struct(self.device)
        td = _t(torch, dtype)
        gen = torch.Generator(device=self.device); gen.manual_seed(seed)
        n3 = g.shape3

        def rnd(shape, lo=0.0, hi=1.0):
            return (torch.rand(shape, generator=gen, device=self.device, dtype=td) * (hi - lo) + lo).contiguous()
        self.u, self.v, self.w = rnd(n3, -1, 1), rnd(n3, -1, 1), rnd(n3, -0.5, 0.5)
        self.w[g.kstart] = 0; self.w[g.kend:] = 0; self.w[:g.kstart] = 0
        self.ut, self.vt, self.wt = rnd(n3, 0, 1e-3), rnd(n3, 0, 1e-3), rnd(n3, 0, 1e-3)
        self.wt[:g.kstart+1] = 0; self.wt[g.kend:] = 0
        self.s, self.st = [], []
        zc = torch.from_numpy(g.z.astype(np.float64)).to(self.device).to(td)
        for _ in range(cfg["nscalars"]):
            th = 300. + 0.003*zc[:, None, None] + rnd(n3, -0.05, 0.05)
            self.s.append(th.contiguous()); self.st.append(rnd(n3, 0, 1e-4))
        self.evisc = torch.zeros(n3, device=self.device, dtype=td)
        self.p = torch.zeros(n3, device=self.device, dtype=td)
        ones = np.ones(g.kcells, dtype=g.np_dtype)
        self.rhoref_h, self.rhorefh_h = ones.copy(), ones.copy()
        self.rhoref, self.rhorefh = torch.from_numpy(ones).to(self.device), torch.from_numpy(ones).to(self.device)
        n2 = g.shape2
        self.surf = {k: rnd(n2, 0, 1e-2) for k in ("u_fluxbot", "u_fluxtop", "v_fluxbot", "v_fluxtop", "s_fluxbot", "s_fluxtop", "dudz", "dvdz")}
        self.surf["dbdz"] = rnd(n2, 0, 1e-4)
        self.surf["z0m"] = torch.full(n2, 0.1, device=self.device, dtype=td)
        self.thref = torch.full((g.kcells,), 300., device=self.device, dtype=td)
        self.work = torch.zeros(16, device=self.device, dtype=torch.float64)
        # Diff_smag2::prepare_device: per-level mixing length table
        self.params = p = capi.MhhDiffParams()
        p.cs, p.tPr, p.surface_model, p.neutral, p.N2, p.th_for_N2, p.grav = 0.23, 1./3., cfg["sm"], 0, None, 0, 9.81
        p.thref = self.thref.data_ptr()
        if cfg["diff"] == DIFF_SMAG2:
            ml = np.zeros(g.kcells, dtype=g.np_dtype)
            capi.check(self.lib.mhh_smag2_mlen0_host(g.host_struct(), p.cs, ml.ctypes.data))
            self.mlen0 = torch.from_numpy(ml).to(self.device)
            p.mlen0 = self.mlen0.data_ptr()
        self.fields = self._fields()
        # Pres::init/set_values/prepare_device
        self.plan = capi.PLAN()
        capi.check(self.lib.mhh_pres_plan_create(g.host_struct(), cfg["pres"], g.dz.ctypes.data, g.dzhi.ctypes.data, g.dzi4.ctypes.data, g.dzhi4.ctypes.data,
                                                 self.rhoref_h.ctypes.data, self.rhorefh_h.ctypes.data, C.byref(self.plan)))
        self._prog = (C.c_void_p * (3 + len(self.s)))(*[t.data_ptr() for t in [self.u, self.v, self.w] + self.s])
        self.cyclic_prognostic()
        torch.cuda.synchronize(self.device)

    def _fields(self):
        f = capi.MhhFields()
        for n in ("u", "v", "w", "ut", "vt", "wt", "evisc", "p", "rhoref", "rhorefh"):
            setattr(f, n, getattr(self, n).data_ptr())
        f.nscalars = len(self.s)
        for n in range(len(self.s)):
            f.s[n], f.st[n], f.svisc[n] = self.s[n].data_ptr(), self.st[n].data_ptr(), self.cfg["visc"]
            f.s_fluxbot[n], f.s_fluxtop[n] = self.surf["s_fluxbot"].data_ptr(), self.surf["s_fluxtop"].data_ptr()
        f.visc = self.cfg["visc"]
        for n in ("u_fluxbot", "u_fluxtop", "v_fluxbot", "v_fluxtop", "dudz", "dvdz", "dbdz", "z0m"):
            setattr(f, n, self.surf[n].data_ptr())
        return f

    @property
    def stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    # -- the operator calls ---------------------------------------------------------------------------
    def cyclic_prognostic(self):
        capi.check(self.lib.mhh_boundary_cyclic_n(self.G, self._prog, len(self._prog), EDGE_BOTH, self.stream))

    def exec_viscosity(self):
        capi.check(self.lib.mhh_diff_exec_viscosity(self.G, self.cfg["diff"], C.byref(self.fields), C.byref(self.params), self.stream))

    def rhs(self):
        capi.check(self.lib.mhh_rhs_exec(self.G, self.cfg["advec"], self.cfg["diff"], C.byref(self.fields), C.byref(self.params), self.stream))

    def rhs_unfused(self):
        capi.check(self.lib.mhh_advec_exec(self.G, self.cfg["advec"], C.byref(self.fields), self.stream))
        capi.check(self.lib.mhh_diff_exec(self.G, self.cfg["diff"], C.byref(self.fields), C.byref(self.params), self.stream))

    def pres(self):
        capi.check(self.lib.mhh_pres_exec(self.plan, self.G, C.byref(self.fields), self.dt, self.stream))

    def step(self):
        """One full RHS + pressure evaluation (the BASELINE metric's unit of work)."""
        self.cyclic_prognostic()
        self.exec_viscosity()
        self.rhs()
        self.pres()

    def divergence(self):
        out = C.c_double(0)
        capi.check(self.lib.mhh_pres_check_divergence(self.G, self.cfg["pres"], C.byref(self.fields), self.work.data_ptr(), C.byref(out), self.stream))
        return out.value

    def cfl(self, dt):
        out = C.c_double(0)
        capi.check(self.lib.mhh_advec_cfl(self.G, self.cfg["advec"], self.u.data_ptr(), self.v.data_ptr(), self.w.data_ptr(), dt, self.work.data_ptr(), C.byref(out), self.stream))
        return out.value

    def close(self):
        if self.plan:
            self.lib.mhh_pres_plan_destroy(self.plan)
            self.plan = None

    # algorithmic bytes per interior cell (SURVEY.md §8d / BASELINE.md §3)
    def alg_bytes_rhs(self):
        s = self.grid.np_dtype.itemsize
        F = 3 + len(self.s)
        return (3*F)*s if self.cfg["diff"] != DIFF_SMAG2 else (F + 1 + 2*F)*s   # pass B only; evisc pass = (F+1)*s

    def alg_bytes_visc(self):
        s = self.grid.np_dtype.itemsize
        return (3 + len(self.s) + 1)*s if self.cfg["diff"] == DIFF_SMAG2 else 0

    def alg_bytes_pres(self):
        return 26*self.grid.np_dtype.itemsize
