"""Host-side driver of one RK sub-step's hot path: the slice of ``Model<TF>::exec`` (src/model.cxx:346-411)
that this package accelerates, expressed as calls into the C ABI.

    boundary->set_prognostic_cyclic_bcs   -> mhh_boundary_cyclic_n (+ N-S neighbour exchange)   (src/model.cxx:346)
    diff->exec_viscosity(thermo)          -> mhh_diff_exec_viscosity (+ N-S exchange of evisc)   (:354)
    advec->exec ; diff->exec              -> mhh_rhs_exec (fused, same bits)                     (:388, :392)
    pres->exec(dt)                        -> mhh_pres_exec, or its slab form around 2 all-to-alls (:411)

One process per GPU. With ``npy`` > 1 the grid is slab-decomposed in y (npx = 1): halos travel as ring
send/recv and the pressure solver's x<->y transposes as all_to_all, both through torch.distributed
(backend "nccl" = RCCL over xGMI). PyTorch provides device memory, streams and the process group only;
every kernel is the hand-written HIP behind the C ABI. Case recipes follow SURVEY.md §8(d).
"""
import ctypes as C
import math
import os

import numpy as np

from . import capi
from .grid import Grid, ADVEC_2, ADVEC_2I5, ADVEC_4, DIFF_2, DIFF_4, DIFF_SMAG2, EDGE_BOTH, EDGE_EW, moser_z

CASES = {
    # name: advec, diff, pres order, spatial order, ghost cells, domain, scalars, surface model
    "taylorgreen": dict(advec=ADVEC_2, diff=DIFF_2, pres=2, order=2, gc=(1, 1, 1), size=(1., 1., 0.5), nscalars=0, sm=0, visc=(8.*math.pi**2*1000.)**-1),
    "drycblles": dict(advec=ADVEC_2I5, diff=DIFF_SMAG2, pres=2, order=2, gc=(3, 3, 1), size=(3200., 3200., 1200.), nscalars=1, sm=1, visc=1e-5),
    # gabls1 (cases/gabls1/gabls1.ini: cs = 0.1, 400 m domain at 32^2 columns, scaled with the column count)
    "gabls1": dict(advec=ADVEC_2I5, diff=DIFF_SMAG2, pres=2, order=2, gc=(3, 3, 1), size=(12800., 12800., 400.), nscalars=1, sm=1, visc=1e-5, cs=0.1),
    "moser600": dict(advec=ADVEC_4, diff=DIFF_4, pres=4, order=4, gc=(3, 3, 3), size=(2*math.pi, math.pi, 2.), nscalars=0, sm=0, visc=1e-5),
}

FIELDS3 = ("u", "v", "w", "ut", "vt", "wt")
SURF = ("u_fluxbot", "u_fluxtop", "v_fluxbot", "v_fluxtop", "s_fluxbot", "s_fluxtop", "dudz", "dvdz", "dbdz", "z0m")


def synthetic_global(case, itot, jtot, ktot, dtype=np.float64, seed=666):
    """Global synthetic fields on the host (interior only; ghosts are filled by the halo code). For tests that
    compare a slab-decomposed run with a single-rank run."""
    cfg = CASES[case]
    rs = np.random.RandomState(seed)
    n3, n2 = (ktot, jtot, itot), (jtot, itot)
    out = {"u": rs.uniform(-1, 1, n3), "v": rs.uniform(-1, 1, n3), "w": rs.uniform(-.5, .5, n3),
           "ut": rs.uniform(0, 1e-3, n3), "vt": rs.uniform(0, 1e-3, n3), "wt": rs.uniform(0, 1e-3, n3)}
    out["w"][0] = 0; out["wt"][0] = 0
    z = (np.arange(ktot) + 0.5) * cfg["size"][2] / ktot
    for n in range(cfg["nscalars"]):
        out["s%d" % n] = 300. + 0.003*z[:, None, None] + rs.uniform(-.05, .05, n3)
        out["st%d" % n] = rs.uniform(0, 1e-4, n3)
    for k in SURF:
        out[k] = rs.uniform(0, 1e-4 if k == "dbdz" else 1e-2, n2)
    out["z0m"][:] = 0.1
    return {k: v.astype(dtype) for k, v in out.items()}


class HotPath:
    """Device-resident fields of ONE rank + the operator calls of one sub-step."""

    def __init__(self, case, itot, jtot, ktot, dtype=np.float64, device="cuda:0", seed=666, dt=1.0,
                 lib=None, npy=1, rank=0, group=None, global_init=None, force_slab=False, slim_halos=True, overlap=None, pres_chunks=None, igc=None):
        import torch
        self.torch = torch
        self.lib = lib if lib is not None else capi.lib()
        self.cfg = cfg = CASES[case]
        self.case, self.dt, self.npy, self.rank, self.group = case, dt, npy, rank, group
        # slab code path (halo pack/unpack, split pressure solve); force_slab runs it on ONE rank with the exchanges
        # degenerated to local copies, which is how the slab kernels are exercised on a single-GPU box
        self.slab = (npy > 1) or force_slab
        # slim_halos: exchange only what the next kernel reads across the slab edge (one row of vt and of p, one direction
        # each) and evaluate evisc on the two adjacent ghost rows locally instead of exchanging it
        self.slim = bool(slim_halos)
        # overlap: work the rows that need no north-south halo while the prognostic halos travel on a second stream
        # (halo_visc_rhs). The default with more than one rank (None = auto; MHH_OVERLAP=0 / overlap=False switch it off,
        # MHH_OVERLAP=1 / overlap=True force it, also on one rank with force_slab): the exchange it hides (25.5 MB each way per
        # rank at 512^3 / 8) costs more on the wire than the ~0.2 ms of smaller edge launches the split adds.
        env = os.environ.get("MHH_OVERLAP")
        if env in ("0", "1"):                        # the environment wins over the argument, both ways
            overlap = (env == "1")
        elif overlap is None:
            overlap = npy > 1
        self.overlap = bool(overlap)
        self._comm_stream = None
        self.device = torch.device(device)
        self.on_gpu = self.device.type == "cuda"
        if self.on_gpu:
            # the C ABI allocates plan buffers with hipMalloc on the CURRENT device and launches on a raw stream handle: make
            # this HotPath's device the current one (cuda:N without a prior torch.cuda.set_device(N) would otherwise put the
            # plan on device 0 and the fields on device N)
            torch.cuda.set_device(self.device)
        # Rehearsal of the N > 1 path on a box with fewer GPUs than ranks: several ranks share one card and talk over
        # gloo, which has no device-side send/recv or all-to-all, so the MESSAGES (never the compute) pass through host
        # copies. With the nccl backend (= RCCL, the product path) the device buffers go to the collective as they are.
        self._host_staged = False
        # MHH_FORCE_COMM=1 (tests): with one rank, still send the halos / transposes / maxima through torch.distributed (to
        # self) instead of the local-copy shortcuts -- exercises the real RCCL calls on a one-GPU box
        self._force_comm = os.environ.get("MHH_FORCE_COMM", "0") == "1"
        # bench.py: device-event pairs around every exchange (tag, start, end) while comm_timing is a list
        self.comm_timing = None
        if self.on_gpu and (npy > 1 or self._force_comm):
            import torch.distributed as dist
            self._host_staged = dist.is_initialized() and dist.get_backend(group) == "gloo"
        if self.slab and cfg["pres"] != 2:
            raise ValueError("slab decomposition implements pres_2 (BASELINE.json multi-GPU configs use pres_2)")
        z = moser_z(ktot, cfg["size"][2]) if case == "moser600" else None
        # igc: ghost cells in x beyond what the schemes need -- what the reference's grid produces when an operator calls
        # Grid::set_minimum_ghost_cells (src/grid.cxx:435-439; src/advec_2i5.cxx:42-45 uses it). igc = 16 with itot = 512 makes
        # rows of 544 cells = 34 whole 128-byte lines with istart on a line (fp64).
        self.grid = g = Grid(itot, jtot, ktot, *cfg["size"], order=cfg["order"], igc=max(cfg["gc"][0], igc or 0), jgc=cfg["gc"][1], kgc=cfg["gc"][2],
                             z=z, dtype=dtype, npy=npy, mpicoordy=rank)
        self.G = g.device_struct(self.device) if self.on_gpu else g.host_struct()
        self.td = td = torch.float64 if np.dtype(dtype) == np.float64 else torch.float32
        n3, n2 = g.shape3, g.shape2
        if global_init is None:
            gen = torch.Generator(device=self.device); gen.manual_seed(seed + 7919*rank)

            def rnd(shape, lo=0.0, hi=1.0):
                return (torch.rand(shape, generator=gen, device=self.device, dtype=td) * (hi - lo) + lo).contiguous()
            self.u, self.v, self.w = rnd(n3, -1, 1), rnd(n3, -1, 1), rnd(n3, -0.5, 0.5)
            self.ut, self.vt, self.wt = rnd(n3, 0, 1e-3), rnd(n3, 0, 1e-3), rnd(n3, 0, 1e-3)
            zc = torch.from_numpy(g.z.astype(np.float64)).to(self.device).to(td)
            self.s = [(300. + 0.003*zc[:, None, None] + rnd(n3, -0.05, 0.05)).contiguous() for _ in range(cfg["nscalars"])]
            self.st = [rnd(n3, 0, 1e-4) for _ in range(cfg["nscalars"])]
            self.surf = {k: rnd(n2, 0, 1e-2) for k in SURF}
            self.surf["dbdz"] = rnd(n2, 0, 1e-4)
            self.surf["z0m"] = torch.full(n2, 0.1, device=self.device, dtype=td)
        else:
            j0 = rank * g.jmax

            def put3(a):
                t = torch.zeros(n3, dtype=td)
                t[g.kstart:g.kend, g.jstart:g.jend, g.istart:g.iend] = torch.from_numpy(np.ascontiguousarray(a[:, j0:j0+g.jmax, :]))
                return t.to(self.device).contiguous()

            def put2(a):
                t = torch.zeros(n2, dtype=td)
                t[g.jstart:g.jend, g.istart:g.iend] = torch.from_numpy(np.ascontiguousarray(a[j0:j0+g.jmax, :]))
                return t.to(self.device).contiguous()
            for n in FIELDS3:
                setattr(self, n, put3(global_init[n]))
            self.s = [put3(global_init["s%d" % n]) for n in range(cfg["nscalars"])]
            self.st = [put3(global_init["st%d" % n]) for n in range(cfg["nscalars"])]
            self.surf = {k: put2(global_init[k]) for k in SURF}
        # solid walls: w and its tendency vanish at kstart and above kend-1
        self.w[:g.kstart+1] = 0; self.w[g.kend:] = 0
        self.wt[:g.kstart+1] = 0; self.wt[g.kend:] = 0
        self.evisc = torch.zeros(n3, device=self.device, dtype=td)
        self.p = torch.zeros(n3, device=self.device, dtype=td)
        ones = np.ones(g.kcells, dtype=g.np_dtype)
        self.rhoref_h, self.rhorefh_h = ones.copy(), ones.copy()
        self.rhoref, self.rhorefh = torch.from_numpy(ones.copy()).to(self.device), torch.from_numpy(ones.copy()).to(self.device)
        self.thref = torch.full((g.kcells,), 300., device=self.device, dtype=td)
        self.work = torch.zeros(16, device=self.device, dtype=torch.float64)
        # Diff_smag2::prepare_device: per-level mixing-length table from the host libm
        self.params = p = capi.MhhDiffParams()
        p.cs, p.tPr, p.surface_model, p.neutral, p.N2, p.th_for_N2, p.grav = cfg.get("cs", 0.23), 1./3., cfg["sm"], 0, None, 0, 9.81
        p.thref = self.thref.data_ptr()
        if cfg["diff"] == DIFF_SMAG2:
            ml = np.zeros(g.kcells, dtype=g.np_dtype)
            self._ok(self.lib.mhh_smag2_mlen0_host(g.host_struct(), p.cs, ml.ctypes.data))
            self.mlen0 = torch.from_numpy(ml).to(self.device)
            p.mlen0 = self.mlen0.data_ptr()
            # Diff_smag2::prepare_device, continued: a horizontally uniform roughness length makes the squared mixing length a
            # per-level table (same bits as the per-cell evaluation; three divisions and a square root per cell less)
            z0 = self.surf["z0m"]
            if bool((z0 == z0.flatten()[0]).all()):
                m2 = np.zeros(g.kcells, dtype=g.np_dtype)
                self._ok(self.lib.mhh_smag2_mlen2_host(g.host_struct(), p.surface_model, p.neutral, ml.ctypes.data, float(z0.flatten()[0]), m2.ctypes.data))
                self.mlen2 = torch.from_numpy(m2).to(self.device)
                p.mlen2 = self.mlen2.data_ptr()
        self.fields = self._fields()
        # Pres::init / set_values / prepare_device
        self.plan = capi.PLAN()
        Gh = g.host_struct()
        if not self.slab:
            self._ok(self.lib.mhh_pres_plan_create(Gh, cfg["pres"], g.dz.ctypes.data, g.dzhi.ctypes.data, g.dzi4.ctypes.data, g.dzhi4.ctypes.data,
                                                   self.rhoref_h.ctypes.data, self.rhorefh_h.ctypes.data, C.byref(self.plan)))
        else:
            self._ok(self.lib.mhh_pres_slab_plan_create(Gh, g.dz.ctypes.data, g.dzhi.ctypes.data, self.rhoref_h.ctypes.data, self.rhorefh_h.ctypes.data, C.byref(self.plan)))
            nx = int(self.lib.mhh_pres_slab_xbuf_elems(self.plan))
            self.xsend = torch.zeros(2*nx, device=self.device, dtype=td)
            self.xrecv = torch.zeros(2*nx, device=self.device, dtype=td)
            self._halo = {}
            # k-slices of the pressure solve: the all-to-all of slice c travels (second stream) while slice c+1 is transformed.
            # Default with more than one rank: 4 slices where ktot allows (MHH_PRES_CHUNKS / pres_chunks override; 1 = unsliced).
            if pres_chunks is None:
                env = os.environ.get("MHH_PRES_CHUNKS")
                pres_chunks = int(env) if env else (4 if npy > 1 else 1)
            pres_chunks = max(1, int(pres_chunks))
            while pres_chunks > 1 and (ktot % pres_chunks or not self.slim):
                pres_chunks -= 1
            self.pres_chunks = pres_chunks
            self._ok(self.lib.mhh_pres_slab_set_chunks(self.plan, pres_chunks))
        self._prog = [self.u, self.v, self.w] + self.s
        self.evisc_local_ghosts = self.slab and self.slim and cfg["diff"] == DIFF_SMAG2 and g.jgc >= 2
        if self.evisc_local_ghosts:
            p.evisc_ghost_rows = 1
            for k in ("dudz", "dvdz", "dbdz", "z0m"):
                self._halo2d(self.surf[k])
        self.cyclic_prognostic()
        self.sync()

    # -- plumbing -----------------------------------------------------------------------------------------
    def _ok(self, rc):
        capi.check(rc, self.lib)

    def sync(self):
        if self.on_gpu:
            self.torch.cuda.synchronize(self.device)

    @property
    def stream(self):
        if not self.on_gpu:
            return C.c_void_p(0)
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

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

    @staticmethod
    def _ptrs(tensors):
        return (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])

    # -- halos ------------------------------------------------------------------------------------------------
    def halo(self, tensors, rows_south=None, rows_north=None):
        """Periodic ghost cells of 3-D fields: east-west wrap on the device, north-south wrap locally (npy == 1) or
        by exchanging rows with the ring neighbours (Boundary_cyclic::exec, src/boundary_cyclic.cxx:116-176).
        rows_south / rows_north: how many of my southernmost / northernmost interior rows travel to the south / north
        neighbour (default jgc each = the reference's full exchange)."""
        arr = self._ptrs(tensors)
        if not self.slab:
            self._ok(self.lib.mhh_boundary_cyclic_n(self.G, arr, len(tensors), EDGE_BOTH, self.stream))
            return
        self._ok(self.lib.mhh_boundary_cyclic_n(self.G, arr, len(tensors), EDGE_EW, self.stream))
        self._exchange_ns(tensors, rows_south, rows_north)

    def _exchange_ns(self, tensors, rows_south=None, rows_north=None):
        """North-south part of halo(): pack, ring exchange (or local swap on one rank), unpack -- on the current stream."""
        arr = self._ptrs(tensors)
        g, nf = self.grid, len(tensors)
        rs = g.jgc if rows_south is None else rows_south
        rn = g.jgc if rows_north is None else rows_north
        key = (nf, rs, rn)
        if key not in self._halo:
            per_row = nf * g.kcells * g.icells
            # one send and one receive buffer, [northbound | southbound]: what I send north arrives as my north neighbour's
            # "from south" part, so with two ranks (north == south) the whole buffer is ONE message pair
            nn, ns = rn * per_row, rs * per_row
            send = self.torch.zeros(max(1, nn + ns), device=self.device, dtype=self.td)
            recv = self.torch.zeros(max(1, nn + ns), device=self.device, dtype=self.td)
            self._halo[key] = [send[nn:nn+ns], send[:nn], recv[:nn], recv[nn:nn+ns], send, recv]   # send south, send north, recv from south, recv from north
        s_south, s_north, r_south, r_north, send, recv = self._halo[key]
        isz = send.element_size()                                   # (an empty view has a null data_ptr: offsets by hand)
        off_s = s_north.numel() * isz
        self._ok(self.lib.mhh_halo_pack_rows(self.G, arr, nf, rs, rn, send.data_ptr() + off_s, send.data_ptr(), self.stream))
        self._ring(rs, rn, s_south, s_north, r_south, r_north, send, recv)
        self._ok(self.lib.mhh_halo_unpack_rows(self.G, arr, nf, rs, rn, recv.data_ptr(), recv.data_ptr() + off_s, self.stream))

    def _timed(self, tag):
        """Context manager: records a pair of events on the current stream around an exchange when comm_timing is on."""
        hp = self

        class _T:
            def __enter__(self_):
                self_.on = hp.comm_timing is not None and hp.on_gpu
                if self_.on:
                    self_.a = hp.torch.cuda.Event(enable_timing=True); self_.b = hp.torch.cuda.Event(enable_timing=True)
                    self_.a.record()

            def __exit__(self_, *exc):
                if self_.on:
                    self_.b.record(); hp.comm_timing.append((tag, self_.a, self_.b))
                return False
        return _T()

    def _ring(self, rs, rn, s_south, s_north, r_south, r_north, send, recv):
        """The message part of _exchange_ns: my northbound rows to the north neighbour, southbound rows to the south one."""
        with self._timed("halo"):
            self._ring_impl(rs, rn, s_south, s_north, r_south, r_north, send, recv)

    def _ring_impl(self, rs, rn, s_south, s_north, r_south, r_north, send, recv):
        import torch.distributed as dist
        if self.npy == 1 and not self._force_comm:          # both neighbours are this rank: the exchange is a local swap
            r_south.copy_(s_north); r_north.copy_(s_south)
        else:
            south, north = (self.rank - 1) % self.npy, (self.rank + 1) % self.npy
            ranks = dist.get_process_group_ranks(self.group) if self.group is not None else list(range(self.npy))
            hs, hr = (send.cpu(), self.torch.empty_like(recv, device="cpu")) if self._host_staged else (send, recv)
            nn, ns = (s_north.numel() if rn else 0), (s_south.numel() if rs else 0)
            if north == south:                                   # two ranks: both halves travel to the same peer
                ops = [dist.P2POp(dist.isend, hs, ranks[north], self.group), dist.P2POp(dist.irecv, hr, ranks[north], self.group)]
            else:
                ops = []
                if rn: ops.append(dist.P2POp(dist.isend, hs[:nn], ranks[north], self.group))
                if rs: ops.append(dist.P2POp(dist.isend, hs[nn:nn+ns], ranks[south], self.group))
                if rn: ops.append(dist.P2POp(dist.irecv, hr[:nn], ranks[south], self.group))
                if rs: ops.append(dist.P2POp(dist.irecv, hr[nn:nn+ns], ranks[north], self.group))
            for w in dist.batch_isend_irecv(ops):
                w.wait()
            if self._host_staged:
                recv.copy_(hr)

    def _halo2d(self, t):
        """One-time periodic ghost cells of a 2-D surface array (Boundary_cyclic::exec_2d, src/boundary_cyclic.cxx:445-500)."""
        g = self.grid
        if self.npy == 1 and not self._force_comm:
            self._ok(self.lib.mhh_boundary_cyclic_2d(self.G, t.data_ptr(), self.stream))
            return
        import torch.distributed as dist
        t[:, :g.igc] = t[:, g.iend-g.igc:g.iend].clone(); t[:, g.iend:] = t[:, g.istart:g.istart+g.igc].clone()
        s_north, s_south = t[g.jend-g.jgc:g.jend].contiguous(), t[g.jstart:g.jstart+g.jgc].contiguous()
        if self._host_staged:
            s_north, s_south = s_north.cpu(), s_south.cpu()
        r_south, r_north = self.torch.empty_like(s_north), self.torch.empty_like(s_south)
        south, north = (self.rank - 1) % self.npy, (self.rank + 1) % self.npy
        ranks = dist.get_process_group_ranks(self.group) if self.group is not None else list(range(self.npy))
        if north == south:                                       # two ranks: one message pair, [northbound | southbound]
            both, got = self.torch.cat([s_north, s_south]), self.torch.cat([r_south, r_north])
            for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, both, ranks[north], self.group), dist.P2POp(dist.irecv, got, ranks[north], self.group)]):
                w.wait()
            r_south, r_north = got[:g.jgc], got[g.jgc:]
        else:
            ops = [dist.P2POp(dist.isend, s_north, ranks[north], self.group), dist.P2POp(dist.isend, s_south, ranks[south], self.group),
                   dist.P2POp(dist.irecv, r_south, ranks[south], self.group), dist.P2POp(dist.irecv, r_north, ranks[north], self.group)]
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        t[:g.jgc] = r_south.to(t.device); t[g.jend:] = r_north.to(t.device)

    def cyclic_prognostic(self):
        self.halo(self._prog)

    # -- the north-south exchange of the prognostic fields hidden behind the rows that do not need it -------------------
    @property
    def can_overlap(self):
        g = self.grid
        return (self.slab and self.overlap and self.evisc_local_ghosts and self.cfg["advec"] == ADVEC_2I5 and self.cfg["diff"] == DIFF_SMAG2
                and len(self.s) == 1 and g.jgc >= 3 and g.jmax >= 12)

    def halo_visc_rhs(self, ev=None):
        """cyclic_prognostic + exec_viscosity + rhs of a slab rank with the halo exchange of u, v, w, th (jgc rows of four
        fields each way: the largest message of the step) travelling on a second stream while the rows that need no
        north-south halo are worked: evisc on rows [jstart+1, jend-1), tendencies on rows [jstart+4, jend-4); then the
        edge rows. Row-wise calls give the bits of the whole-slab calls (tests/test_slab_gloo.py)."""
        g, lib, torch = self.grid, self.lib, self.torch
        adv, dif = self.cfg["advec"], self.cfg["diff"]
        F, P = C.byref(self.fields), C.byref(self.params)
        self._ok(lib.mhh_boundary_cyclic_n(self.G, self._ptrs(self._prog), len(self._prog), EDGE_EW, self.stream))
        if self.on_gpu:
            main = torch.cuda.current_stream(self.device)
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream(self.device)
                self._ev = [torch.cuda.Event(), torch.cuda.Event()]
            self._ev[0].record(main)
            self._comm_stream.wait_event(self._ev[0])
            with torch.cuda.stream(self._comm_stream):
                self._exchange_ns(self._prog)
                self._ev[1].record(self._comm_stream)
        else:
            self._exchange_ns(self._prog)
        ja, jb = g.jstart + 4, g.jend - 4
        if ev is not None:
            ev[2].record()
        self._ok(lib.mhh_diff_exec_viscosity_rows(self.G, dif, F, P, g.jstart + 1, g.jend - 1, self.stream))
        if ev is not None:
            ev[0].record()
        self._ok(lib.mhh_rhs_exec_rows(self.G, adv, dif, F, P, ja, jb, self.stream))
        if ev is not None:
            ev[1].record()
        self.rhs_rows_timed = jb - ja
        if self.on_gpu:
            torch.cuda.current_stream(self.device).wait_event(self._ev[1])
        # both edge strips in one launch each
        self._ok(lib.mhh_diff_exec_viscosity_rows2(self.G, dif, F, P, g.jstart - 1, g.jstart + 1, g.jend - 1, g.jend + 1, self.stream))
        self._ok(lib.mhh_rhs_exec_rows2(self.G, adv, dif, F, P, g.jstart, ja, jb, g.jend, self.stream))

    # -- the operator calls -----------------------------------------------------------------------------------
    def exec_viscosity(self):
        self._ok(self.lib.mhh_diff_exec_viscosity(self.G, self.cfg["diff"], C.byref(self.fields), C.byref(self.params), self.stream))
        if self.slab and self.cfg["diff"] == DIFF_SMAG2 and not self.evisc_local_ghosts:
            self.halo([self.evisc])

    def rhs(self):
        self._ok(self.lib.mhh_rhs_exec(self.G, self.cfg["advec"], self.cfg["diff"], C.byref(self.fields), C.byref(self.params), self.stream))

    def rhs_unfused(self):
        self._ok(self.lib.mhh_advec_exec(self.G, self.cfg["advec"], C.byref(self.fields), self.stream))
        self._ok(self.lib.mhh_diff_exec(self.G, self.cfg["diff"], C.byref(self.fields), C.byref(self.params), self.stream))

    def pres(self):
        if not self.slab:
            self._ok(self.lib.mhh_pres_exec(self.plan, self.G, C.byref(self.fields), self.dt, self.stream))
            return
        import torch.distributed as dist
        lib, st = self.lib, self.stream
        if self.slim: self.halo([self.vt], rows_south=1, rows_north=0)      # only vt[j+1] at the north edge is read (pres_2.cxx:181,193)
        else:         self.halo([self.vt])
        F = C.byref(self.fields)
        # x stages with the transforms in LDS where the plan has them (needs the slim halos: p's y halo is the one row): input + x
        # transform write the send buffer, x transform + p + output read the receive buffer -- no packed array, no pack / unpack
        lds_x = self.slim and lib.mhh_pres_slab_has_lds(self.plan) == 1
        packed = lib.mhh_pres_slab_packed(self.plan)
        if not lds_x:
            self._ok(lib.mhh_pres_input_packed(self.G, 2, F, self.dt, packed, st))
        if self.pres_chunks > 1:
            self._pres_sliced(packed, lds_x)
            return
        if lds_x:
            self._ok(lib.mhh_pres_slab_lds_fwd(self.plan, self.G, F, self.dt, self.xsend.data_ptr(), 0, st))
        else:
            self._ok(lib.mhh_pres_fwd_x_pack(self.plan, self.G, packed, self.xsend.data_ptr(), st))
        self._transpose()                                                    # Transpose::exec_xy
        if lds_x:       # the y transforms read / write the transposes' buffers directly, around the Thomas sweeps
            self._ok(lib.mhh_pres_slab_lds_fwd_y(self.plan, self.G, self.xrecv.data_ptr(), 0, self.stream))
            self._ok(lib.mhh_pres_solve_y(self.plan, self.G, self.stream))
            self._ok(lib.mhh_pres_slab_lds_bwd_y(self.plan, self.G, self.xsend.data_ptr(), 0, self.stream))
        else:
            self._ok(lib.mhh_pres_fwd_y_solve_bwd_y(self.plan, self.G, self.xrecv.data_ptr(), self.xsend.data_ptr(), self.stream))
        self._transpose()                                                    # Transpose::exec_yx
        if lds_x:
            self._ok(lib.mhh_pres_slab_lds_bwd(self.plan, self.G, self.xrecv.data_ptr(), F, 0, self.stream))
            self.halo([self.p], rows_south=0, rows_north=1)
            self._ok(lib.mhh_pres_output_south_row(self.G, F, self.stream))
            return
        if self.slim:
            # unpack + Pres_2::output in one kernel for everything but vt on the southernmost row, whose p[j-1] arrives with the
            # one-row halo of p (pres_2.cxx:383-385)
            self._ok(lib.mhh_pres_bwd_x_unpack_output(self.plan, self.G, self.xrecv.data_ptr(), C.byref(self.fields), self.stream))
            self.halo([self.p], rows_south=0, rows_north=1)
            self._ok(lib.mhh_pres_output_south_row(self.G, C.byref(self.fields), self.stream))
        else:
            self._ok(lib.mhh_pres_bwd_x_unpack(self.plan, self.G, self.xrecv.data_ptr(), C.byref(self.fields), self.stream))
            self.halo([self.p])
            self._ok(lib.mhh_pres_output_order(self.G, 2, C.byref(self.fields), self.stream))

    def pres_rk(self, rkorder, substep, dt):
        """pres->exec(self.dt) followed by timeloop.exec() for u, v, w (src/model.cxx:411,484) with the sub-step applied in the
        kernel that stores the corrected tendencies (mhh_pres_exec_rk); on a slab: pres() + three mhh_rk_substep calls."""
        if not self.slab:
            self._ok(self.lib.mhh_pres_exec_rk(self.plan, self.G, C.byref(self.fields), self.dt, rkorder, substep, dt, self.stream))
            return
        self.pres()
        for a, at in ((self.u, self.ut), (self.v, self.vt), (self.w, self.wt)):
            self._ok(self.lib.mhh_rk_substep(self.G, rkorder, substep, dt, a.data_ptr(), at.data_ptr(), self.stream))

    def _pres_sliced(self, packed, lds_x=False):
        """Pres_2::exec after the input stage, in k-slices: per slice x transform + pack, all-to-all on the exchange stream while
        the next slice is transformed, y transform as each slice arrives; Thomas sweeps over all levels; the same on the way back.
        Same kernels per plane as the unsliced path (tests/test_slab_gloo.py compares the two)."""
        lib, torch, n = self.lib, self.torch, self.pres_chunks
        seg = self.xsend.numel() // n
        F = C.byref(self.fields)
        two_streams = self.on_gpu and not self._host_staged and (self.npy > 1 or self._force_comm)
        if two_streams:
            main = torch.cuda.current_stream(self.device)
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream(self.device)
                self._ev = [torch.cuda.Event(), torch.cuda.Event()]
            if not hasattr(self, "_sl_ev"):
                self._sl_ev = [[torch.cuda.Event() for _ in range(n)] for _ in range(4)]

        def exchange(c, ready, done):
            """all-to-all of slice c: on the exchange stream once `ready` (recorded on the main stream) has passed"""
            a, b = self.xsend[c*seg:(c+1)*seg], self.xrecv[c*seg:(c+1)*seg]
            if not two_streams:
                self._transpose(a, b)
                return
            ready.record(main)
            self._comm_stream.wait_event(ready)
            with torch.cuda.stream(self._comm_stream):
                self._transpose(a, b)
                done.record(self._comm_stream)

        for c in range(n):
            if lds_x: self._ok(lib.mhh_pres_slab_lds_fwd(self.plan, self.G, F, self.dt, self.xsend.data_ptr(), c, self.stream))
            else:     self._ok(lib.mhh_pres_fwd_x_pack_chunk(self.plan, self.G, packed, self.xsend.data_ptr(), c, self.stream))
            exchange(c, *( (self._sl_ev[0][c], self._sl_ev[1][c]) if two_streams else (None, None) ))
        for c in range(n):
            if two_streams:
                main.wait_event(self._sl_ev[1][c])
            if lds_x: self._ok(lib.mhh_pres_slab_lds_fwd_y(self.plan, self.G, self.xrecv.data_ptr(), c, self.stream))
            else:     self._ok(lib.mhh_pres_fwd_y_chunk(self.plan, self.G, self.xrecv.data_ptr(), c, self.stream))
        self._ok(lib.mhh_pres_solve_y(self.plan, self.G, self.stream))
        for c in range(n):
            if lds_x: self._ok(lib.mhh_pres_slab_lds_bwd_y(self.plan, self.G, self.xsend.data_ptr(), c, self.stream))
            else:     self._ok(lib.mhh_pres_bwd_y_chunk(self.plan, self.G, self.xsend.data_ptr(), c, self.stream))
            exchange(c, *( (self._sl_ev[2][c], self._sl_ev[3][c]) if two_streams else (None, None) ))
        for c in range(n):
            if two_streams:
                main.wait_event(self._sl_ev[3][c])
            if lds_x: self._ok(lib.mhh_pres_slab_lds_bwd(self.plan, self.G, self.xrecv.data_ptr(), F, c, self.stream))
            else:     self._ok(lib.mhh_pres_bwd_x_chunk(self.plan, self.G, self.xrecv.data_ptr(), c, self.stream))
        if not lds_x:
            self._ok(lib.mhh_pres_unpack_output_slab(self.plan, self.G, F, self.stream))
        self.halo([self.p], rows_south=0, rows_north=1)
        self._ok(lib.mhh_pres_output_south_row(self.G, F, self.stream))

    def _transpose(self, send=None, recv=None):
        """x<->y transpose of the spectral pressure: one equal-split all-to-all (RCCL over xGMI)."""
        if send is not None:
            return self._transpose_buffers(send, recv)
        return self._transpose_buffers(self.xsend, self.xrecv)

    def _transpose_buffers(self, xsend, xrecv):
        if self.npy == 1 and not self._force_comm:
            xrecv.copy_(xsend)
            return
        import torch.distributed as dist
        if self._host_staged:
            hs = xsend.cpu(); hr = self.torch.empty_like(hs)
            dist.all_to_all_single(hr, hs, group=self.group)
            xrecv.copy_(hr)
            return
        with self._timed("transpose"):
            dist.all_to_all_single(xrecv, xsend, group=self.group)

    def step(self):
        """One full RHS + pressure evaluation (the BASELINE metric's unit of work)."""
        if self.can_overlap:
            self.halo_visc_rhs()
        else:
            self.cyclic_prognostic()
            self.exec_viscosity()
            self.rhs()
        self.pres()

    def capture_step(self):
        """One step recorded as a hipGraph (single GPU): every entry point of the library only enqueues work on the stream it is
        given -- no allocation, no synchronisation, no host round trip inside a step -- so the dozen launches of a step (cyclic
        fills, exec_viscosity, fused RHS, the pressure kernels or rocFFT's) replay as ONE graph launch. Matters where a step is
        short (256^3: 1.6 ms, 64^3: 0.1 ms); returns the graph, `graph.replay()` runs a step on the same fields."""
        assert self.on_gpu and not self.slab, "graph capture: single-GPU path"
        torch = self.torch
        self.step(); self.sync()                       # everything created lazily (rocFFT work areas) exists before the capture
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            self.step()
        return graph

    # -- reductions (local max, then MAX over ranks: Master::max, src/master_parallel.cxx:233-266) --------------
    def _allmax(self, v):
        if self.npy == 1 and not self._force_comm:
            return v
        import torch.distributed as dist
        t = self.torch.tensor([v], device=("cpu" if self._host_staged else self.device), dtype=self.torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return float(t.item())

    def divergence(self):
        out = C.c_double(0)
        self._ok(self.lib.mhh_pres_check_divergence(self.G, self.cfg["pres"], C.byref(self.fields), self.work.data_ptr(), C.byref(out), self.stream))
        return self._allmax(out.value)

    def projected_divergence(self):
        """(max |Pres::input| of the fields as they stand, max |Pres::input| of u, v, w alone), each without its horizontal mean per
        level, MAX over ranks: Pres::input is the
        divergence of u/dt + ut the solve has just removed (src/pres_2.cxx:156-196, src/pres_4.cxx:256-317; its halo and ghost-level
        side effects are the ones the next Pres::exec would apply anyway), the second number the scale to read the first against.
        A self-check of the (distributed) solve for bench.py and the tests, outside any timed region."""
        torch, g = self.torch, self.grid
        buf = torch.empty(g.imax*g.jmax*g.kmax, device=self.device, dtype=self.td)

        def residual(f, real):
            if self.slab:
                if real:
                    self.halo([self.vt], rows_south=1, rows_north=0)
                self._ok(self.lib.mhh_pres_input_packed(self.G, self.cfg["pres"], C.byref(f), self.dt, buf.data_ptr(), self.stream))
            else:
                self._ok(self.lib.mhh_pres_input(self.plan, self.G, C.byref(f), self.dt, buf.data_ptr(), self.stream))
            self.sync()
            # without the horizontal mean of each level: the mode (0, 0) has a boundary condition of its own on top (Dirichlet:
            # src/pres_2.cxx:311-316, src/pres_4.cxx:421-446) and pres_4 does not enforce its equation on the top levels -- with
            # synthetic fields, whose w has a horizontal mean, that mode is left with a remainder that says nothing about the solve
            r = buf.view(g.kmax, g.jmax, g.imax)
            lev = r.sum(dim=(1, 2), dtype=torch.float64)
            if self.npy > 1 or self._force_comm:
                import torch.distributed as dist
                t = lev.cpu() if self._host_staged else lev
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
                lev = t.to(self.device)
            mean = (lev / float(g.itot * g.jtot)).to(self.td)
            return self._allmax(float((r - mean[:, None, None]).abs().max().item()))
        d1 = residual(self.fields, True)
        zero = torch.zeros_like(self.ut)
        f0 = self._fields()
        f0.ut = f0.vt = f0.wt = zero.data_ptr()
        d0 = residual(f0, False)
        return d1, d0

    def cfl(self, dt):
        out = C.c_double(0)
        self._ok(self.lib.mhh_advec_cfl(self.G, self.cfg["advec"], self.u.data_ptr(), self.v.data_ptr(), self.w.data_ptr(), dt, self.work.data_ptr(), C.byref(out), self.stream))
        return self._allmax(out.value)

    # -- restart files in the reference's layout (microhh_amd/fieldio.py; src/field3d_io.cxx:54-230) ------------------
    def _restart_fields(self):
        names = [("u", self.u), ("v", self.v), ("w", self.w)]
        names += [("th" if n == 0 else "s%d" % n, t) for n, t in enumerate(self.s)]
        return names

    def save(self, path, iteration=0):
        """Write the prognostic fields as `name.NNNNNNN`; slab ranks write their rows of the one global file. Rank 0 creates
        every file at its full size, all ranks meet at a barrier, each writes its rows, and a second barrier ends the call
        (a reader on another rank then sees complete files)."""
        from . import fieldio
        names = [(fieldio.field_filename(path, name, iteration), t) for name, t in self._restart_fields()]
        if self.rank == 0:
            fieldio.save_grid(path, self.grid, jtot=self.grid.jmax * self.npy)
            if self.npy > 1:
                for fn, _ in names:
                    fieldio.prepare_global_file(fn, self.grid, self.npy)
        self._barrier()
        for fn, t in names:
            fieldio.save_field3d(fn, t.detach().cpu().numpy(), self.grid, rank=self.rank, npy=self.npy)
        self._barrier()

    def _barrier(self):
        if self.npy > 1:
            import torch.distributed as dist
            if dist.is_initialized():
                dist.barrier(self.group)

    def load(self, path, iteration=0):
        """Read the prognostic fields' interiors back (ghost cells are refreshed by the next cyclic_prognostic())."""
        from . import fieldio
        for name, t in self._restart_fields():
            a = fieldio.load_field3d(fieldio.field_filename(path, name, iteration), self.grid, rank=self.rank, npy=self.npy,
                                     out=t.detach().cpu().numpy().copy())
            t.copy_(self.torch.from_numpy(a.reshape(-1)).to(t.device).reshape(t.shape))

    def close(self):
        if self.plan:
            (self.lib.mhh_pres_slab_plan_destroy if self.slab else self.lib.mhh_pres_plan_destroy)(self.plan)
            self.plan = None

    # -- algorithmic bytes per interior cell (SURVEY.md §8d / BASELINE.md §3) -------------------------------------
    def alg_bytes_rhs(self):
        s = self.grid.np_dtype.itemsize
        F = 3 + len(self.s)
        return (3*F)*s if self.cfg["diff"] != DIFF_SMAG2 else (F + 1 + 2*F)*s   # smag2: pass B (tendencies); pass A below

    def alg_bytes_visc(self):
        s = self.grid.np_dtype.itemsize
        return (3 + len(self.s) + 1)*s if self.cfg["diff"] == DIFF_SMAG2 else 0

    def alg_bytes_pres(self):
        return 26*self.grid.np_dtype.itemsize
