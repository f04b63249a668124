"""bench.py's contract (one JSON line on rank 0, the keys the driver reads, max-over-ranks timing, N ranks through
torch.distributed.run on 127.0.0.1) rehearsed on the CPU: gloo instead of RCCL, the test-only emulation of the kernels
instead of the GPU library. The numbers mean nothing; the control flow -- slab decomposition, halo ring, all-to-alls,
reductions, JSON assembly -- is the one the GPU run takes."""
import json
import os
import subprocess
import sys

import pytest

import backends as B
import common as cm

KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline")


def _run(world, extra=()):
    B.get("emul")                                        # makes sure the emulation library is built
    env = dict(os.environ, MHH_LIB=os.path.join(cm.ROOT, "tests", "emul", "libmhh_emul.so"), OMP_NUM_THREADS="1")
    args = ["--gpus", str(world), "--steps", "2", "--warmup", "1", "--device", "cpu", "--workload", "rehearsal", "--no-cpu-baseline"] + list(extra)
    if world == 1:
        cmd = [sys.executable, os.path.join(cm.ROOT, "bench.py")] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(29600 + world + os.getpid() % 300), os.path.join(cm.ROOT, "bench.py")] + args
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=cm.ROOT, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                       # exactly one JSON line, from rank 0
    return json.loads(lines[0])


@pytest.mark.parametrize("world", [1, 2])
def test_bench_json_line(world):
    d = _run(world)
    for k in KEYS:
        assert k in d, k
    assert d["n_gpus"] == world and d["steps"] == 2 and d["warmup"] == 1
    assert d["value"] > 0 and d["ms_per_step"] > 0 and d["higher_is_better"] is True and d["scaling"] == "strong"
    assert d["vs_baseline"] is None and d["dtype"] == "f64" and "REHEARSAL" in d["data"]
    assert d["config"]["grid"] == [16, 24, 10] and d["config"]["decomposition"].endswith("npy=%d" % world)
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"]/r["peak"]) < 1e-12
    c = d["self_check"]             # the (distributed) solve did its job on the benchmark's own fields: div(u/dt + ut) gone
    assert c["max_abs_pres_input_of_u_over_dt"] > 1e-3 and c["ratio"] < 1e-9, c


def test_bench_overlapped_path_two_ranks():
    d = _run(2, extra=[])                                 # default with more than one rank: the halo exchange overlaps the interior rows
    assert d["config"]["halo_overlap"] is True and d["value"] > 0 and d["self_check"]["ratio"] < 1e-9
    env_was = os.environ.get("MHH_OVERLAP")
    os.environ["MHH_OVERLAP"] = "0"
    try:
        d = _run(2)
    finally:
        if env_was is None:
            os.environ.pop("MHH_OVERLAP", None)
        else:
            os.environ["MHH_OVERLAP"] = env_was
    assert d["config"]["halo_overlap"] is False and d["value"] > 0
