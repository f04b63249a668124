"""The N > 1 path (slab decomposition in y, halo ring exchange, pressure solve around two all-to-alls) on the CPU:
world_size 2 and 4 with the gloo backend, kernels = the library's own sources executed by the test-only HIP
stand-in (tests/emul). Each run is compared with the single-rank run on the same global synthetic fields:
RHS tendencies and evisc bit-exact (same stencils, same halos), pressure-corrected tendencies to 1e-10
(the transform is split x / y instead of 2-D)."""
import os
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import backends as B  # noqa: E402
from microhh_amd.model import HotPath, synthetic_global  # noqa: E402

GRID = (16, 32, 10)        # jmax = 16 / 8 rows per rank: whole strips of eight, so the slim-halo runs take the LDS x stages of the pressure solve


def _interior(hp, t):
    g = hp.grid
    return t[g.kstart:g.kend, g.jstart:g.jend, g.istart:g.iend].numpy().copy()


def _run(hp, out, overlapped=False):
    if overlapped:
        assert hp.can_overlap
        hp.halo_visc_rhs()               # interior rows while the halos travel, then the edge rows
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


def _worker(rank, world, port, tmp, slim, lds_x=True):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ["MHH_PRES_SLAB_LDS"] = "1" if lds_x else "0"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lib = B.get("emul").lib
        hp = HotPath("drycblles", *GRID, device="cpu", lib=lib, npy=world, rank=rank, global_init=synthetic_global("drycblles", *GRID),
                     slim_halos=slim, overlap=(slim and world == 2))
        assert hp.evisc_local_ghosts == slim
        assert lib.mhh_pres_slab_has_lds(hp.plan) == (1 if lds_x else 0)
        out = {}
        _run(hp, out, overlapped=(slim and world == 2))
        np.savez(os.path.join(tmp, "rank%d.npz" % rank), **out)
        hp.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,slim,lds_x", [(2, True, True), (4, True, True), (2, False, True), (2, True, False)], ids=["2-slim", "4-slim", "2-full-halos", "2-slim-staged-x"])
def test_slab_matches_single_rank(world, slim, lds_x):
    """slim: one-row vt / p exchanges and evisc evaluated on the adjacent ghost rows; full: the reference's jgc-row
    exchanges of vt, p and evisc. Both must reproduce the single-rank bits. The slim runs solve the pressure with the x stages in
    LDS writing / reading the all-to-all buffers (mhh_pres_slab_lds_fwd / _bwd), "staged-x" and the full-halo run with the rocFFT
    x stages (input | x transform | pack and the reverse)."""
    lib = B.get("emul").lib
    ref = {}
    hp = HotPath("drycblles", *GRID, device="cpu", lib=lib, global_init=synthetic_global("drycblles", *GRID))
    _run(hp, ref)
    hp.close()
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_worker, args=(world, 29500 + world + 7*int(slim) + 13*int(lds_x) + os.getpid() % 1000, tmp, slim, lds_x), nprocs=world, join=True)
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
            assert abs(float(p["div"]) - float(ref["div"])) <= 1e-12 * abs(float(ref["div"]))


def _save_worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import time
        lib = B.get("emul").lib
        hp = HotPath("drycblles", *GRID, device="cpu", lib=lib, npy=world, rank=rank, global_init=synthetic_global("drycblles", *GRID))
        if rank == 0:
            time.sleep(0.5)              # rank 0 (which creates the files) arrives LAST: the others must wait for it
        hp.save(tmp, iteration=3)
        hp.save(tmp, iteration=3)        # again over existing files of the right size: nobody's rows may be wiped
        hp.close()
    finally:
        dist.destroy_process_group()


def test_slab_ranks_save_restart_files_concurrently():
    """HotPath.save from two ranks at once (rank 0 creates and sizes every file, barrier, all write, barrier): a
    stale file of the wrong size is in the way, rank 0 is late, and the result must equal the single-rank files byte for byte."""
    lib = B.get("emul").lib
    gi = synthetic_global("drycblles", *GRID)
    with tempfile.TemporaryDirectory() as one, tempfile.TemporaryDirectory() as two:
        hp = HotPath("drycblles", *GRID, device="cpu", lib=lib, global_init=gi)
        hp.save(one, iteration=3); names = [n for n, _ in hp._restart_fields()]; hp.close()
        with open(os.path.join(two, "u.0000003"), "wb") as f:
            f.write(b"stale")
        mp.spawn(_save_worker, args=(2, 29700 + os.getpid() % 1000, two), nprocs=2, join=True)
        for n in names + ["grid"]:
            fn = "%s.%07d" % (n, 0 if n == "grid" else 3)
            assert open(os.path.join(one, fn), "rb").read() == open(os.path.join(two, fn), "rb").read(), fn


def test_sliced_pressure_solve_equals_unsliced_on_one_rank():
    """The k-sliced form of the slab pressure solve (mhh_pres_*_chunk: slice c of the all-to-all can travel while slice c+1 is
    transformed) against the unsliced calls, one rank, emulated kernels: every plane takes the same transforms -> same bits."""
    lib = B.get("emul").lib
    gi = synthetic_global("drycblles", *GRID)
    out = {}
    for n in (1, 2, 5):
        hp = HotPath("drycblles", *GRID, device="cpu", lib=lib, global_init=gi, force_slab=True, pres_chunks=n)
        assert hp.pres_chunks == n
        hp.cyclic_prognostic(); hp.exec_viscosity(); hp.rhs(); hp.pres()
        out[n] = {k: _interior(hp, getattr(hp, k)) for k in ("p", "ut", "vt", "wt")}
        hp.close()
    for n in (2, 5):
        for k in ("p", "ut", "vt", "wt"):
            assert np.array_equal(out[n][k], out[1][k]), (n, k)
