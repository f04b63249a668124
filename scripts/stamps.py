"""Diagnostic: per-segment cycle shares of the marching kernel (variant build with -DMHH_MARCH_STAMPS)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MHH_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "microhh_amd", "variants", "libmhh_hip_stamps.so")
import torch
from microhh_amd.model import HotPath
from microhh_amd import capi
hp = HotPath("drycblles", 512, 512, 512)
lib = capi.lib()
out = (C.c_ulonglong * 8)()
hp.exec_viscosity(); hp.rhs(); torch.cuda.synchronize()
lib.mhh_debug_march_stamps(out)
for _ in range(3): hp.rhs()
torch.cuda.synchronize()
lib.mhh_debug_march_stamps(out)
v = [x for x in out]; tot = sum(v)
names = ["loop-top/prev shift", "prefetch issue", "top-face quantities", "tendency update", "barrier 1 (wait)", "LDS tile stores", "barrier 2 (wait)", "-"]
for n, x in zip(names, v):
    print("%-24s %6.2f %%" % (n, 100.0*x/max(tot, 1)))
