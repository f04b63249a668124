"""Does one sub-step (halo + exec_viscosity + fused RHS + pressure with rocFFT) capture into a HIP graph and replay?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from microhh_amd.model import HotPath

for case, shape in (("taylorgreen", (64, 64, 64)), ("drycblles", (256, 256, 256)), ("drycblles", (512, 512, 512))):
    hp = HotPath(case, *shape)
    for _ in range(3): hp.step()
    torch.cuda.synchronize()
    def timeit(fn, n=50):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return 1e3*(time.perf_counter()-t0)/n
    eager = timeit(hp.step)
    ref = [x.clone() for x in (hp.ut, hp.vt, hp.wt, hp.p)]
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    try:
        with torch.cuda.stream(s):
            hp.step()                       # warm-up on the side stream
            with torch.cuda.graph(g, stream=s):
                hp.step()
        torch.cuda.current_stream().wait_stream(s)
        graph = timeit(g.replay)
        print(case, shape, "eager %.3f ms  graph %.3f ms" % (eager, graph), flush=True)
    except Exception as e:
        print(case, shape, "eager %.3f ms  capture failed: %r" % (eager, e), flush=True)
    hp.close()
