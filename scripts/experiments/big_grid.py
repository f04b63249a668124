"""One MI355X holds much more than the BASELINE grids: a 1024 x 1024 x 512 fp64 drycblles step (0.54 G cells, 4.3 GB per field)
as a capacity check -- timing, and agreement of the fused pass with the two operator calls."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from microhh_amd.model import HotPath

shape = tuple(int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (1024, 1024, 512)
hp = HotPath("drycblles", *shape)
print("allocated %.1f GB" % (torch.cuda.memory_allocated()/1e9), flush=True)
hp.cyclic_prognostic(); hp.exec_viscosity()
keep = [t.clone() for t in (hp.ut, hp.vt, hp.wt, hp.st[0])]
hp.rhs(); hp.sync(); fused = [t.clone() for t in (hp.ut, hp.vt, hp.wt, hp.st[0])]
for t, k in zip((hp.ut, hp.vt, hp.wt, hp.st[0]), keep): t.copy_(k)
hp.rhs_unfused(); hp.sync()
for a, b, n in zip(fused, (hp.ut, hp.vt, hp.wt, hp.st[0]), "uvws"):
    assert torch.equal(a, b), n
del fused, keep
for _ in range(2): hp.step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): hp.step()
torch.cuda.synchronize(); ms = 1e3*(time.perf_counter()-t0)/5
n = shape[0]*shape[1]*shape[2]
print(shape, "%.2f ms/step  %.2f Gcell/s  fused == two operator calls" % (ms, n/ms/1e6), flush=True)
hp.close()
