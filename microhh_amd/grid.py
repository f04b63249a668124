"""Host-side grid bookkeeping: the index bundle and stretched-z metrics the hot path consumes.

Mirror of the reference's ``Grid<TF>::calculate`` (src/grid.cxx:237-368) and ``Grid_data``
(include/grid.h:49-135).  The metrics are *inputs* of the hot path (SURVEY.md §2 row 11,
"CONSUMED"): they are computed once on the host, in double precision, and narrowed to the
build's float type; nothing here is accelerated.

The ctypes structure ``MhhGrid`` is the C-ABI ``mhh_grid`` of include/mhh_hip.h.
"""
import ctypes as C
import numpy as np

MHH_F64, MHH_F32 = 0, 1
EDGE_EW, EDGE_NS, EDGE_BOTH = 0, 1, 2
ADVEC_2, ADVEC_2I5, ADVEC_4 = 2, 25, 4
ADVEC_2I4, ADVEC_2I62, ADVEC_2I53, ADVEC_4M = 24, 262, 253, 41
DIFF_2, DIFF_4, DIFF_SMAG2 = 2, 4, 22
MAX_SCALARS = 8


class MhhGrid(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "itot", "jtot", "ktot", "imax", "jmax", "kmax", "igc", "jgc", "kgc",
        "icells", "jcells", "ijcells", "kcells", "istart", "jstart", "kstart",
        "iend", "jend", "kend", "dtype", "npx", "npy", "mpicoordx", "mpicoordy")] + [
        ("ncells", C.c_longlong),
        ("xsize", C.c_double), ("ysize", C.c_double), ("zsize", C.c_double),
        ("dx", C.c_double), ("dy", C.c_double)] + [
        (n, C.c_void_p) for n in ("z", "zh", "dz", "dzh", "dzi", "dzhi", "dzi4", "dzhi4")]


METRICS = ("z", "zh", "dz", "dzh", "dzi", "dzhi", "dzi4", "dzhi4")

# 4th-order weights, include/finite_difference.h:62-95
CI = (-1./16., 9./16., 9./16., -1./16.)
BI = (5./16., 15./16., -5./16., 1./16.)
TI = (1./16., -5./16., 15./16., 5./16.)
CG = (1./24., -27./24., 27./24., -1./24.)
BG = (-23./24., 21./24., 3./24., -1./24.)
TG = (1./24., -3./24., -21./24., 23./24.)
DHUGE = 1.e30


def _w4(w, a, b, c, d):
    return w[0]*a + w[1]*b + w[2]*c + w[3]*d


def uniform_z(ktot, zsize):
    """z_k = (k+1/2) zsize/ktot (cases/drycblles/drycblles_input.py:15-20)."""
    dz = zsize / ktot
    return np.linspace(0.5*dz, zsize-0.5*dz, ktot)


def moser_z(ktot, zsize, alpha=0.967):
    """tanh-stretched channel grid (cases/moser600/moser600_input.py:20-23)."""
    k = np.arange(ktot)
    eta = -1. + 2.*((k+1)-0.5) / ktot
    return zsize / (2.*alpha) * np.tanh(eta*0.5*(np.log(1.+alpha) - np.log(1.-alpha))) + 0.5*zsize


class Grid:
    """Index bundle + metrics for one rank's sub-domain (slab in y: npx == 1)."""

    def __init__(self, itot, jtot, ktot, xsize, ysize, zsize, order=2, igc=None, jgc=None, kgc=None,
                 z=None, dtype=np.float64, npy=1, mpicoordy=0):
        if order not in (2, 4):
            raise ValueError("swspatialorder must be 2 or 4")
        if jtot % npy or itot % npy:
            raise ValueError("itot and jtot must be divisible by npy (src/grid.cxx:111-131)")
        self.order = order
        self.np_dtype = np.dtype(dtype)
        self.dtype = MHH_F64 if self.np_dtype == np.float64 else MHH_F32
        d = 1 if order == 2 else 3
        self.igc = d if igc is None else igc
        self.jgc = d if jgc is None else jgc
        self.kgc = d if kgc is None else kgc
        self.itot, self.jtot, self.ktot = itot, jtot, ktot
        self.npx, self.npy, self.mpicoordx, self.mpicoordy = 1, npy, 0, mpicoordy
        self.imax, self.jmax, self.kmax = itot, jtot // npy, ktot
        self.iblock, self.jblock = itot // npy, jtot      # pencils after the x<->y transpose
        self.icells = self.imax + 2*self.igc
        self.jcells = self.jmax + 2*self.jgc
        self.kcells = self.kmax + 2*self.kgc
        self.ijcells = self.icells * self.jcells
        self.ncells = self.ijcells * self.kcells
        self.istart, self.jstart, self.kstart = self.igc, self.jgc, self.kgc
        self.iend, self.jend, self.kend = self.istart+self.imax, self.jstart+self.jmax, self.kstart+self.kmax
        tf = self.np_dtype.type
        self.xsize, self.ysize, self.zsize = float(tf(xsize)), float(tf(ysize)), float(tf(zsize))
        self.dx = float(tf(tf(xsize) / tf(itot)))
        self.dy = float(tf(tf(ysize) / tf(jtot)))
        if jtot > 1 and self.jmax < self.jgc:
            raise ValueError("jmax must be >= jgc (src/grid.cxx:420)")
        zin = uniform_z(ktot, zsize) if z is None else np.asarray(z, dtype=np.float64)
        if zin.shape != (ktot,):
            raise ValueError("z must have ktot entries")
        self._calculate(zin)
        self._dev = None

    # -- Grid::calculate, src/grid.cxx:237-368 ---------------------------------------------------
    def _calculate(self, zin):
        kc, ks, ke = self.kcells, self.kstart, self.kend
        z = np.zeros(kc); zh = np.zeros(kc)
        dz = np.zeros(kc); dzh = np.zeros(kc); dzi = np.zeros(kc); dzhi = np.zeros(kc)
        dzi4 = np.zeros(kc); dzhi4 = np.zeros(kc)
        z[ks:ke] = zin
        zs = self.zsize
        if self.order == 2:
            z[ks-1] = -z[ks]
            z[ke] = 2.*zs - z[ke-1]
            for k in range(ks+1, ke):
                zh[k] = 0.5*(z[k-1]+z[k])
            zh[ks] = 0.; zh[ke] = zs
            for k in range(1, kc):
                dzh[k] = z[k] - z[k-1]; dzhi[k] = 1./dzh[k]
            dzh[ks-1] = dzh[ks+1]; dzhi[ks-1] = dzhi[ks+1]
            for k in range(1, kc-1):
                dz[k] = zh[k+1] - zh[k]
                dzi[k] = 1./dz[k] if dz[k] != 0. else 0.
            dz[ks-1] = dz[ks]; dzi[ks-1] = dzi[ks]
            dz[ke] = dz[ke-1]; dzi[ke] = dzi[ke-1]
        else:
            z[ks-1] = -2.*z[ks] + (1./3.)*z[ks+1]
            z[ks-2] = -9.*z[ks] + 2.*z[ks+1]
            z[ke] = (8./3.)*zs - 2.*z[ke-1] + (1./3.)*z[ke-2]
            z[ke+1] = 8.*zs - 9.*z[ke-1] + 2.*z[ke-2]
            z[ks-3] = DHUGE; z[ke+2] = DHUGE
            zh[ks] = 0.
            for k in range(ks+1, ke):
                zh[k] = _w4(CI, z[k-2], z[k-1], z[k], z[k+1])
            zh[ke] = zs
            zh[ks-1] = _w4(BI, z[ks-2], z[ks-1], z[ks], z[ks+1])
            zh[ke+1] = _w4(TI, z[ke-2], z[ke-1], z[ke], z[ke+1])
            for k in range(1, kc):
                dzh[k] = z[k] - z[k-1]; dzhi[k] = 1./dzh[k]
            dzh[ks-3] = dzh[ks+3]; dzhi[ks-3] = dzhi[ks+3]
            for k in range(1, kc-1):
                dz[k] = zh[k+1] - zh[k]
                dzi[k] = 1./dz[k] if dz[k] != 0. else 0.
            dz[ks-3] = dz[ks+2]; dzi[ks-3] = dzi[ks+2]
            dz[ke+2] = dz[ke-3]; dzi[ke+2] = dzi[ke-3]
            for k in range(ks, ke):
                dzi4[k] = 1./_w4(CG, zh[k-1], zh[k], zh[k+1], zh[k+2])
                dzhi4[k] = 1./_w4(CG, z[k-2], z[k-1], z[k], z[k+1])
            dzhi4[ke] = 1./_w4(CG, z[ke-2], z[ke-1], z[ke], z[ke+1])
            dzi4[ks-1] = 1./_w4(BG, zh[ks-1], zh[ks], zh[ks+1], zh[ks+2])
            dzhi4[ks-1] = 1./_w4(BG, z[ks-2], z[ks-1], z[ks], z[ks+1])
            dzi4[ke] = 1./_w4(TG, zh[ke-2], zh[ke-1], zh[ke], zh[ke+1])
            dzhi4[ke+1] = 1./_w4(TG, z[ke-2], z[ke-1], z[ke], z[ke+1])
            dzi4[ks-2] = DHUGE; dzi4[ks-3] = DHUGE; dzi4[ke+1] = DHUGE; dzi4[ke+2] = DHUGE
        t = self.np_dtype
        with np.errstate(over="ignore"):
            self.z, self.zh = z.astype(t), zh.astype(t)
            self.dz, self.dzh = dz.astype(t), dzh.astype(t)
            self.dzi, self.dzhi = dzi.astype(t), dzhi.astype(t)
            self.dzi4, self.dzhi4 = dzi4.astype(t), dzhi4.astype(t)

    # -- ABI structs --------------------------------------------------------------------------------
    def _fill(self, ptrs):
        g = MhhGrid()
        for n in ("itot", "jtot", "ktot", "imax", "jmax", "kmax", "igc", "jgc", "kgc", "icells", "jcells",
                  "ijcells", "kcells", "istart", "jstart", "kstart", "iend", "jend", "kend", "dtype",
                  "npx", "npy", "mpicoordx", "mpicoordy", "ncells", "xsize", "ysize", "zsize", "dx", "dy"):
            setattr(g, n, getattr(self, n))
        for n in METRICS:
            setattr(g, n, ptrs[n])
        return g

    def host_struct(self):
        """pointer to an mhh_grid whose metric pointers are HOST numpy buffers (oracle / _ref / plan creation)."""
        g = self._fill({n: getattr(self, n).ctypes.data for n in METRICS})
        pg = C.pointer(g)
        pg._keep = self
        return pg

    def device_struct(self, device="cuda"):
        """pointer to an mhh_grid whose metric pointers are DEVICE buffers (torch tensors own the memory)."""
        import torch
        if self._dev is None:
            self._dev = {n: torch.from_numpy(getattr(self, n)).to(device) for n in METRICS}
        g = self._fill({n: self._dev[n].data_ptr() for n in METRICS})
        pg = C.pointer(g)
        pg._keep = self
        return pg

    # shapes
    @property
    def shape3(self):
        return (self.kcells, self.jcells, self.icells)

    @property
    def shape2(self):
        return (self.jcells, self.icells)

    @property
    def interior(self):
        return (slice(self.kstart, self.kend), slice(self.jstart, self.jend), slice(self.istart, self.iend))
