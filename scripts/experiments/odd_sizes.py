import sys, time, torch
sys.path.insert(0, "/root/repo")
from microhh_amd.model import HotPath
for shape in ((384, 384, 384), (480, 360, 200), (768, 768, 128)):
    hp = HotPath("drycblles", *shape)
    for _ in range(3): hp.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): hp.step()
    torch.cuda.synchronize(); ms = 1e3*(time.perf_counter()-t0)/10
    n = shape[0]*shape[1]*shape[2]
    print(shape, "%.3f ms/step  %.2f Gcell/s  div %.3e" % (ms, n/ms/1e6, hp.divergence()), flush=True)
    hp.close()
