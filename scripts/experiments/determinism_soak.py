"""Soak test for ordering hazards (LDS-DMA rings, deferred stores, prefetched tendencies): the same sub-step from the same
inputs, many times, must give the same bits every time. Prints one line per configuration."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from microhh_amd.model import HotPath

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for case, shape, dtype in (("drycblles", (512, 512, 512), np.float64), ("drycblles", (256, 256, 256), np.float64), ("moser600", (512, 256, 256), np.float64),
                           ("drycblles", (518, 250, 130), np.float64), ("gabls1", (1024, 512, 128), np.float32)):
    hp = HotPath(case, *shape, dtype=dtype)
    state = [hp.ut, hp.vt, hp.wt, hp.p, hp.evisc] + list(hp.st)
    init = [t.clone() for t in state]
    def run():
        for t, k in zip(state, init): t.copy_(k)
        hp.step()
    run(); torch.cuda.synchronize()
    ref = [t.clone() for t in state]
    bad = 0; t0 = time.perf_counter()
    for n in range(reps):
        run()
        if not all(torch.equal(a, b) for a, b in zip(state, ref)): bad += 1
    torch.cuda.synchronize()
    print("%-10s %-16s %s  %d repetitions, %d differ  (%.1f s)" % (case, shape, np.dtype(dtype).name, reps, bad, time.perf_counter()-t0), flush=True)
    hp.close()
    assert bad == 0
print("soak ok")
