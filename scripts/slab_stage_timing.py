"""Per-rank compute time of the slab-decomposed pipeline at its REAL per-rank shape (e.g. rank 0 of 8 on 512^3), on
one GPU: the exchanges are skipped (buffers keep whatever they hold), only the kernels either side are timed."""
import ctypes as C, os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from microhh_amd import capi
from microhh_amd.model import HotPath

npy = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
lib = capi.lib()
hp = HotPath("drycblles", n, n, n, npy=npy, rank=0, group=None, global_init=None) if False else None
# build the rank-0 object without a process group: construct with npy ranks but never call the exchanges
class NoComm(HotPath):
    def _ring(self, *a):
        pass
    def _halo2d(self, t):
        pass
    def _transpose(self, *a):
        pass
hp = NoComm("drycblles", n, n, n, npy=npy, rank=0)
def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/reps
lib, st = hp.lib, hp.stream
packed = lib.mhh_pres_slab_packed(hp.plan)
f = C.byref(hp.fields)
res = {
 "cyclic_prognostic(4 fields)": timeit(hp.cyclic_prognostic),
 "exec_viscosity+halo": timeit(hp.exec_viscosity),
 "rhs": timeit(hp.rhs),
 "halo(vt, 1 row south)": timeit(lambda: hp.halo([hp.vt], rows_south=1, rows_north=0)),
 "pres_input": timeit(lambda: lib.mhh_pres_input_packed(hp.G, 2, f, 1.0, packed, st)),
 "fwd_x_pack": timeit(lambda: lib.mhh_pres_fwd_x_pack(hp.plan, hp.G, packed, hp.xsend.data_ptr(), st)),
 "fwd_y_solve_bwd_y": timeit(lambda: lib.mhh_pres_fwd_y_solve_bwd_y(hp.plan, hp.G, hp.xrecv.data_ptr(), hp.xsend.data_ptr(), st)),
 "bwd_x_unpack_output (fused)": timeit(lambda: lib.mhh_pres_bwd_x_unpack_output(hp.plan, hp.G, hp.xrecv.data_ptr(), f, st)),
 "halo(p, 1 row north)": timeit(lambda: hp.halo([hp.p], rows_south=0, rows_north=1)),
 "pres_output south row": timeit(lambda: lib.mhh_pres_output_south_row(hp.G, f, st)),
 "LDS x stage 1: input + x transform -> send buffer": timeit(lambda: lib.mhh_pres_slab_lds_fwd(hp.plan, hp.G, f, 1.0, hp.xsend.data_ptr(), 0, st)) if lib.mhh_pres_slab_has_lds(hp.plan) and hp.pres_chunks == 1 else float("nan"),
 "LDS x stage 3: receive buffer -> x transform + p + output": timeit(lambda: lib.mhh_pres_slab_lds_bwd(hp.plan, hp.G, hp.xrecv.data_ptr(), f, 0, st)) if lib.mhh_pres_slab_has_lds(hp.plan) and hp.pres_chunks == 1 else float("nan"),
 "LDS y stage: y transform in | Thomas | y transform out": timeit(lambda: (lib.mhh_pres_slab_lds_fwd_y(hp.plan, hp.G, hp.xrecv.data_ptr(), 0, st), lib.mhh_pres_solve_y(hp.plan, hp.G, st), lib.mhh_pres_slab_lds_bwd_y(hp.plan, hp.G, hp.xsend.data_ptr(), 0, st))) if lib.mhh_pres_slab_has_lds(hp.plan) and hp.pres_chunks == 1 else float("nan"),
 "(two-kernel form) bwd_x_unpack": timeit(lambda: lib.mhh_pres_bwd_x_unpack(hp.plan, hp.G, hp.xrecv.data_ptr(), f, st)),
 "(two-kernel form) pres_output": timeit(lambda: lib.mhh_pres_output_order(hp.G, 2, f, st)),
 "full step (no comm)": timeit(hp.step),
}
for k, v in res.items(): print("%-58s %8.3f ms" % (k, v))
print("x stages in LDS: %s; k-slices of the transposes: %d" % ("yes" if lib.mhh_pres_slab_has_lds(hp.plan) else "no (MHH_PRES_SLAB_LDS=0 or no such form)", hp.pres_chunks))
print("all-to-all volume per rank per direction: %.1f MB" % (hp.xsend.numel()*8/1e6))
