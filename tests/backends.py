"""Test backends for the parity suite.

* ``hip``  -- the product: microhh_amd/libmhh_hip.so on a real MI355X, torch CUDA tensors own the memory and the
              kernels run on torch's current stream. Tests using it carry @pytest.mark.gpu.
* ``emul`` -- the SAME kernel sources compiled for the CPU with the test-only HIP stand-in (tests/emul); numpy
              arrays own the memory. It exists to debug kernel logic without a GPU and is never used by
              the package itself.
"""
import ctypes as C
import os
import subprocess

import numpy as np

import common as cm
from microhh_amd import capi
from microhh_amd.grid import METRICS


class EmulBackend:
    name = "emul"

    def __init__(self):
        d = os.path.join(cm.ROOT, "tests", "emul")
        subprocess.run(["make", "-s", "-C", d], check=True)
        self.lib = capi.bind(C.CDLL(os.path.join(d, "libmhh_emul.so")))
        self.stream = C.c_void_p(0)

    def arr(self, a):
        return None if a is None else np.ascontiguousarray(a).copy()

    def zeros(self, shape, dtype):
        return np.zeros(shape, dtype=dtype)

    def ptr(self, a):
        return cm.ptr(a)

    def host(self, a):
        return np.array(a, copy=True)

    def grid(self, g):
        return g.host_struct()

    def view(self, p, shape, dtype):
        """An array over `device` memory the library owns (address p)."""
        ct = C.c_double if np.dtype(dtype) == np.float64 else C.c_float
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(ct)), shape=shape)

    def sync(self):
        pass


class HipBackend:
    name = "hip"

    def __init__(self):
        import torch
        self.torch = torch
        assert torch.cuda.is_available(), "gpu tests need a GPU"
        self.lib = capi.lib()                      # raises if the HIP library is missing: no fallback
        self.dev = torch.device("cuda:0")

    @property
    def stream(self):
        return C.c_void_p(self.torch.cuda.current_stream().cuda_stream)

    def arr(self, a):
        return None if a is None else self.torch.from_numpy(np.ascontiguousarray(a)).to(self.dev)

    def zeros(self, shape, dtype):
        return self.torch.zeros(shape, dtype=self.torch.float64 if np.dtype(dtype) == np.float64 else self.torch.float32, device=self.dev)

    def ptr(self, a):
        return C.c_void_p(0) if a is None else C.c_void_p(a.data_ptr())

    def host(self, a):
        return a.detach().cpu().numpy().copy()

    def grid(self, g):
        return g.device_struct(self.dev)

    def view(self, p, shape, dtype):
        """A copy (device to device) of memory the library owns (address p)."""
        out = self.zeros(shape, dtype)
        self.torch.cuda.synchronize()
        hip = C.CDLL("libamdhip64.so")
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        assert hip.hipMemcpy(out.data_ptr(), p, out.numel()*out.element_size(), 3) == 0
        return out

    def sync(self):
        self.torch.cuda.synchronize()


_cache = {}


def get(name):
    if name not in _cache:
        _cache[name] = EmulBackend() if name == "emul" else HipBackend()
    return _cache[name]


def ok(be, rc):
    capi.check(rc, be.lib)


class DevCase:
    """Device-resident copy of a common.Case plus the C-ABI structs pointing at it."""

    def __init__(self, be, case, diff=None):
        self.be, self.case, self.g = be, case, case.grid
        self.G = be.grid(self.g)
        names = ["u", "v", "w", "ut", "vt", "wt", "evisc", "N2", "dudz", "dvdz", "dbdz", "z0m", "u_fluxbot", "u_fluxtop",
                 "v_fluxbot", "v_fluxtop", "s_fluxbot", "s_fluxtop", "p", "rhoref", "rhorefh"]
        for n in names:
            setattr(self, n, be.arr(getattr(case, n)))
        self.s = [be.arr(a) for a in case.s]
        self.st = [be.arr(a) for a in case.st]
        self.work = be.zeros(16, np.float64)

    def fields(self, visc=1e-5, svisc=1e-5):
        be = self.be
        f = capi.MhhFields()
        for n in ("u", "v", "w", "ut", "vt", "wt", "evisc", "p", "rhoref", "rhorefh", "u_fluxbot", "u_fluxtop", "v_fluxbot", "v_fluxtop",
                  "dudz", "dvdz", "dbdz", "z0m"):
            setattr(f, n, be.ptr(getattr(self, n)).value)
        f.nscalars = len(self.s)
        for n in range(len(self.s)):
            f.s[n] = be.ptr(self.s[n]).value
            f.st[n] = be.ptr(self.st[n]).value
            f.svisc[n] = svisc
            f.s_fluxbot[n] = be.ptr(self.s_fluxbot).value
            f.s_fluxtop[n] = be.ptr(self.s_fluxtop).value
        f.visc = visc
        return f


def mlen0(be, g, cs):
    """Per-level Smagorinsky length table: host pow through the ABI helper, then uploaded."""
    out = np.zeros(g.kcells, dtype=g.np_dtype)
    ok(be, be.lib.mhh_smag2_mlen0_host(g.host_struct(), cs, cm.ptr(out)))
    return be.arr(out)
