"""The torch.distributed calls of the N > 1 path against real RCCL with a one-rank communicator (the box has one GPU):
dtype / view / API plumbing of batch_isend_irecv (to self), all_to_all_single, all_reduce(MAX), barrier."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29733")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
for dt in (torch.float64, torch.float32):
    send = torch.arange(1000, device="cuda", dtype=dt); recv = torch.zeros_like(send)
    nn = 600
    ops = [dist.P2POp(dist.isend, send[:nn], 0), dist.P2POp(dist.isend, send[nn:], 0),
           dist.P2POp(dist.irecv, recv[:nn], 0), dist.P2POp(dist.irecv, recv[nn:], 0)]
    for w in dist.batch_isend_irecv(ops): w.wait()
    torch.cuda.synchronize(); assert torch.equal(send, recv), "p2p to self"
    ops = [dist.P2POp(dist.isend, send, 0), dist.P2POp(dist.irecv, recv, 0)]
    recv.zero_()
    for w in dist.batch_isend_irecv(ops): w.wait()
    torch.cuda.synchronize(); assert torch.equal(send, recv), "single pair"
    a = torch.randn(4096, device="cuda", dtype=dt); b = torch.empty_like(a)
    dist.all_to_all_single(b, a); torch.cuda.synchronize(); assert torch.equal(a, b), "all_to_all_single"
t = torch.tensor([3.25], device="cuda", dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX); assert float(t) == 3.25
dist.barrier(); torch.cuda.synchronize()
print("rccl world-1 probe ok:", torch.cuda.nccl.version() if hasattr(torch.cuda, "nccl") else "")
dist.destroy_process_group()
