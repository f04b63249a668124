"""Soak of the slab code path with its exchanges going through real RCCL (one-rank communicator, MHH_FORCE_COMM=1): the same
sub-step from the same inputs, many times, plain and with the overlapped halo path -- stream-ordering races between the
kernels and RCCL's stream would show as differing bits."""
import os, sys, time
os.environ["MHH_FORCE_COMM"] = "1"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29777")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.distributed as dist
from microhh_amd.model import HotPath

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
for shape, overlap in (((512, 64, 512), False), ((512, 64, 512), True), ((256, 256, 256), False), ((128, 64, 32), True)):
    hp = HotPath("drycblles", *shape, npy=1, rank=0, force_slab=True, overlap=overlap)
    assert hp._force_comm and (hp.can_overlap == overlap)
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
    print("%-16s overlap=%-5s %d repetitions through RCCL, %d differ (%.1f s)" % (shape, overlap, reps, bad, time.perf_counter()-t0), flush=True)
    hp.close()
    assert bad == 0
dist.destroy_process_group()
print("rccl soak ok")
