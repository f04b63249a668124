"""The N > 1 path with the real HIP kernels: 2 and 4 ranks share the one GPU of the test box (one process per rank, as in
production), messages travel over gloo through host copies (RCCL refuses two ranks on one device; gloo has no device
send/recv). What this pins on hardware that the gloo/emulation tests cannot: the slab kernels launched with npy > 1
(rank-dependent mode offsets, padded x-mode blocks, one-row halos, evisc ghost rows, row-window launches of the marching
kernels) against the single-rank run on the same global fields -- RHS tendencies and evisc bit-exact, pressure-corrected
tendencies to 1e-10."""
import os
import sys
import tempfile

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu
GRID = (128, 64, 32)


def _interior(hp, t):
    g = hp.grid
    return t[g.kstart:g.kend, g.jstart:g.jend, g.istart:g.iend].cpu().numpy().copy()


def _run(hp, out, overlapped=False):
    if overlapped:
        assert hp.can_overlap
        hp.halo_visc_rhs()
    else:
        hp.cyclic_prognostic()
        hp.exec_viscosity()
    out["evisc"] = _interior(hp, hp.evisc)
    if not overlapped:
        hp.rhs()
    for n in ("ut", "vt", "wt"):
        out["rhs_" + n] = _interior(hp, getattr(hp, n))
    out["rhs_st"] = _interior(hp, hp.st[0])
    hp.pres()
    for n in ("ut", "vt", "wt", "p"):
        out[n] = _interior(hp, getattr(hp, n))
    out["div"] = np.array(hp.divergence())
    out["cfl"] = np.array(hp.cfl(0.5))


def _worker(rank, world, port, tmp, overlap):
    import torch
    import torch.distributed as dist
    from microhh_amd.model import HotPath, synthetic_global
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        hp = HotPath("drycblles", *GRID, device="cuda:0", npy=world, rank=rank, global_init=synthetic_global("drycblles", *GRID), overlap=overlap)
        assert hp._host_staged and hp.evisc_local_ghosts
        out = {}
        _run(hp, out, overlapped=overlap)
        np.savez(os.path.join(tmp, "rank%d.npz" % rank), **out)
        hp.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,overlap", [(2, False), (4, False), (2, True)], ids=["2", "4", "2-overlap"])
def test_slab_ranks_on_one_gpu_match_single_rank(world, overlap):
    import torch
    import torch.multiprocessing as mp
    from microhh_amd.model import HotPath, synthetic_global
    ref = {}
    hp = HotPath("drycblles", *GRID, device="cuda:0", global_init=synthetic_global("drycblles", *GRID))
    _run(hp, ref)
    hp.close()
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_worker, args=(world, 29700 + world + 11*int(overlap) + os.getpid() % 1000, tmp, overlap), nprocs=world, join=True)
        parts = [np.load(os.path.join(tmp, "rank%d.npz" % r)) for r in range(world)]
        for key in ("evisc", "rhs_ut", "rhs_vt", "rhs_wt", "rhs_st"):
            got = np.concatenate([p[key] for p in parts], axis=1)
            assert np.array_equal(got, ref[key]), key
        for key in ("ut", "vt", "wt", "p"):
            got = np.concatenate([p[key] for p in parts], axis=1)
            scale = np.abs(ref[key]).max()
            assert np.abs(got - ref[key]).max() <= 1e-10 * scale, (key, np.abs(got - ref[key]).max() / scale)
        for p in parts:
            assert float(p["cfl"]) == float(ref["cfl"])
            assert abs(float(p["div"]) - float(ref["div"])) <= 1e-12 * abs(float(ref["div"])) + 1e-18


def _rccl_worker(rank, port, tmp, overlap, chunks):
    import torch
    import torch.distributed as dist
    from microhh_amd.model import HotPath, synthetic_global
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ["MHH_FORCE_COMM"] = "1"
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        hp = HotPath("drycblles", *GRID, device="cuda:0", npy=1, rank=0, force_slab=True, global_init=synthetic_global("drycblles", *GRID), overlap=overlap, pres_chunks=chunks)
        assert hp.pres_chunks == chunks
        assert hp._force_comm and not hp._host_staged and hp.evisc_local_ghosts
        out = {}
        _run(hp, out, overlapped=overlap)
        np.savez(os.path.join(tmp, "rank0.npz"), **out)
        hp.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("overlap,chunks", [(False, 1), (True, 1), (True, 4)], ids=["plain", "overlap", "overlap-sliced-transposes"])
def test_slab_code_path_through_real_rccl_on_one_rank(overlap, chunks):
    """The slab code path with its exchanges going through RCCL itself (nccl backend, one-rank communicator, MHH_FORCE_COMM=1:
    halos as batch_isend_irecv to self, the transposes as all_to_all_single -- whole or in four k-slices on the exchange
    stream while the next slice is transformed --, maxima as all_reduce) -- device buffers handed
    to the collectives as in production, stream ordering between the kernels and RCCL included. Same bits as the plain run."""
    import torch.multiprocessing as mp
    from microhh_amd.model import HotPath, synthetic_global
    ref = {}
    hp = HotPath("drycblles", *GRID, device="cuda:0", global_init=synthetic_global("drycblles", *GRID))
    _run(hp, ref)
    hp.close()
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_rccl_worker, args=(29900 + int(overlap) + 2*chunks + os.getpid() % 1000, tmp, overlap, chunks), nprocs=1, join=True)
        got = np.load(os.path.join(tmp, "rank0.npz"))
        for key in ("evisc", "rhs_ut", "rhs_vt", "rhs_wt", "rhs_st"):
            assert np.array_equal(got[key], ref[key]), key
        for key in ("ut", "vt", "wt", "p"):
            scale = np.abs(ref[key]).max()
            assert np.abs(got[key] - ref[key]).max() <= 1e-10 * scale, (key, np.abs(got[key] - ref[key]).max() / scale)
        assert float(got["cfl"]) == float(ref["cfl"])


def test_bench_line_under_rccl_is_one_json_line_with_comm_times():
    """bench.py with its exchanges through real RCCL (one rank, MHH_FORCE_COMM=1, --force-slab): RCCL's banner must not reach
    stdout (exactly one line, the JSON), and the line carries the per-exchange stream times of an N > 1 run."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MHH_FORCE_COMM="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29950 + os.getpid() % 40))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "rehearsal", "--force-slab", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, env=env, cwd=root, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["config"]["decomposition"].endswith("npy=1")
    assert d["comm"]["exchanges_per_step"] == 5 and d["comm"]["halo_ms_per_step"] > 0 and d["comm"]["transpose_ms_per_step"] > 0
