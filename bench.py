#!/usr/bin/env python3
"""bench.py -- throughput of the per-sub-step RHS + pressure hot path (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" = one full evaluation of  cyclic halo -> Diff::exec_viscosity -> Advec::exec + Diff::exec (fused) ->
Pres::exec  on synthetic drycblles-shaped fields already resident in HBM. value = interior cells of the whole
job / max-over-ranks time per step. Strong scaling: the same global grid is slab-decomposed in y over N ranks.
Prints ONE JSON line (rank 0).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)

WORKLOADS = {
    "drycblles512": ("drycblles", 512, 512, 512, "drycblles 512^3 fp64, advec_2i5 + diff_smag2 + pres_2 (BASELINE.json configs[3] grid on N GPUs)"),
    "drycblles256": ("drycblles", 256, 256, 256, "drycblles 256^3 fp64, advec_2i5 + diff_smag2 + pres_2 (BASELINE.json configs[1])"),
    "moser600": ("moser600", 512, 256, 256, "moser600 512x256x256 fp64, advec_4 + diff_4 + pres_4 (BASELINE.json configs[2])"),
    "slab8of512": ("drycblles", 512, 64, 512, "one rank's share (512x64x512) of drycblles 512^3 on 8 GPUs, run alone: per-rank compute estimate"),
    "rehearsal": ("drycblles", 16, 24, 10, "tiny drycblles grid: contract rehearsal on the CPU (--device cpu), not a measurement"),
    "taylorgreen64": ("taylorgreen", 64, 64, 64, "taylorgreen 64^3 fp64, advec_2 + diff_2 + pres_2 (BASELINE.json configs[0])"),
    # BASELINE.json configs[4]: fp32, RHS only (exec_viscosity + advec_2i5 + diff_smag2; no pressure solve in that config)
    "gabls1_1024": ("gabls1", 1024, 1024, 256, "gabls1 1024x1024x256 fp32, advec_2i5 + diff_smag2, RHS only (BASELINE.json configs[4] grid on N GPUs)"),
    "gabls1_slab8": ("gabls1", 1024, 128, 256, "one rank's share (1024x128x256) of gabls1 1024x1024x256 fp32 on 8 GPUs, run alone, RHS only"),
}
FP32_RHS_ONLY = ("gabls1_1024", "gabls1_slab8")


def cpu_baseline(case, sample=(256, 256, 256), reps=3, with_pres=True):
    """CPU baseline on the GPU box's host cores, ONE thread (the reference CPU path has no intra-rank threading): the
    reference's own stencil translation units compiled in place (oracle/_ref/libmhhref_perf.so: src/advec_*.cxx,
    src/diff_*.cxx at -O3 -march=native -DNDEBUG, the reference's flags) for Advec::exec, Diff::exec and calc_strain2,
    and the oracle port (same flags) for what cannot be built from the reference here: the cyclic fills, N2 + calc_evisc
    (src/grid.cxx needs netcdf.h) and Pres::exec (fftw3.h; the port's FFT is its own radix-2 code, not FFTW).
    Reported beside the GPU number; never part of the measured product path."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import common as cm
    from common import ptr, dbl
    from microhh_amd.model import CASES
    cfg = CASES[case]
    it, jt, kt = sample
    g = cm.Grid(it, jt, kt, *cfg["size"], order=cfg["order"], igc=cfg["gc"][0], jgc=cfg["gc"][1], kgc=cfg["gc"][2],
                z=(cm.moser_z(kt, cfg["size"][2]) if case == "moser600" else None))
    c = cm.Case(g, nscalars=max(cfg["nscalars"], 0), rho="one", periodic=True)
    O = cm.oracle(perf=True); G = g.host_struct()
    R = cm.ref(perf=True)                      # None where neither the reference nor its prebuilt library is present
    thref = np.full(g.kcells, 300.)
    n2 = np.zeros(g.shape3); pk = np.zeros((kt, jt, it))
    a = (ptr(c.u), ptr(c.v), ptr(c.w), ptr(c.rhoref), ptr(c.rhorefh))
    adv, dif, sm = cfg["advec"], cfg["diff"], cfg["sm"]
    ns = cfg["nscalars"]
    refname = {2: "ref_advec_2", 25: "ref_advec_2i5", 4: "ref_advec_4"}.get(adv)
    use_ref = R is not None and refname is not None
    tt = {"ref": 0.0, "port": 0.0}

    def timed(kind, fn, *args):
        t0 = time.perf_counter(); fn(*args); tt[kind] += time.perf_counter() - t0

    def step():
        for f in [c.u, c.v, c.w] + c.s[:ns]:
            timed("port", O.orc_boundary_cyclic, G, ptr(f), 2)
        if dif == 22:
            if use_ref: timed("ref", R.ref_smag2_strain2, G, sm, ptr(c.evisc), ptr(c.u), ptr(c.v), ptr(c.w), ptr(c.dudz), ptr(c.dvdz))
            else:       timed("port", O.orc_smag2_strain2, G, sm, ptr(c.evisc), ptr(c.u), ptr(c.v), ptr(c.w), ptr(c.dudz), ptr(c.dvdz))
            timed("port", O.orc_calc_N2, G, ptr(n2), ptr(c.s[0]), ptr(thref), dbl(9.81))
            timed("port", O.orc_smag2_evisc, G, sm, ptr(c.evisc), ptr(n2), ptr(c.dbdz), ptr(c.z0m), dbl(0.23), dbl(1./3.))
            timed("port", O.orc_boundary_cyclic, G, ptr(c.evisc), 2)
        if use_ref:
            fn = getattr(R, refname)
            for comp, t in ((0, c.ut), (1, c.vt), (2, c.wt)):
                timed("ref", fn, G, comp, ptr(t), None, *a)
            for n in range(ns):
                timed("ref", fn, G, 3, ptr(c.st[n]), ptr(c.s[n]), *a)
        else:
            timed("port", O.orc_advec_u, G, adv, ptr(c.ut), *a); timed("port", O.orc_advec_v, G, adv, ptr(c.vt), *a); timed("port", O.orc_advec_w, G, adv, ptr(c.wt), *a)
            for n in range(ns):
                timed("port", O.orc_advec_s, G, adv, ptr(c.st[n]), ptr(c.s[n]), *a)
        if dif == 22:
            if use_ref:
                for comp, t, fb, ft in ((0, c.ut, c.u_fluxbot, c.u_fluxtop), (1, c.vt, c.v_fluxbot, c.v_fluxtop), (2, c.wt, None, None)):
                    timed("ref", R.ref_smag2_diff_uvw, G, comp, sm, ptr(t), ptr(c.u), ptr(c.v), ptr(c.w), ptr(c.evisc), ptr(fb), ptr(ft), ptr(c.rhoref), ptr(c.rhorefh), dbl(1e-5))
                for n in range(ns):
                    timed("ref", R.ref_smag2_diff_c, G, sm, ptr(c.st[n]), ptr(c.s[n]), ptr(c.evisc), ptr(c.s_fluxbot), ptr(c.s_fluxtop), ptr(c.rhoref), ptr(c.rhorefh), dbl(1./3.), dbl(1e-5))
            else:
                timed("port", O.orc_smag2_diff_u, G, sm, ptr(c.ut), ptr(c.u), ptr(c.v), ptr(c.w), ptr(c.evisc), ptr(c.u_fluxbot), ptr(c.u_fluxtop), ptr(c.rhoref), ptr(c.rhorefh), dbl(1e-5))
                timed("port", O.orc_smag2_diff_v, G, sm, ptr(c.vt), ptr(c.u), ptr(c.v), ptr(c.w), ptr(c.evisc), ptr(c.v_fluxbot), ptr(c.v_fluxtop), ptr(c.rhoref), ptr(c.rhorefh), dbl(1e-5))
                timed("port", O.orc_smag2_diff_w, G, ptr(c.wt), ptr(c.u), ptr(c.v), ptr(c.w), ptr(c.evisc), ptr(c.rhoref), ptr(c.rhorefh), dbl(1e-5))
                for n in range(ns):
                    timed("port", O.orc_smag2_diff_c, G, sm, ptr(c.st[n]), ptr(c.s[n]), ptr(c.evisc), ptr(c.s_fluxbot), ptr(c.s_fluxtop), ptr(c.rhoref), ptr(c.rhorefh), dbl(1./3.), dbl(1e-5))
        else:
            o = 2 if dif == 2 else 4
            if use_ref:
                fn = getattr(R, "ref_diff_%d" % o)
                timed("ref", fn, G, 0, ptr(c.ut), ptr(c.u), dbl(1e-5)); timed("ref", fn, G, 0, ptr(c.vt), ptr(c.v), dbl(1e-5)); timed("ref", fn, G, 1, ptr(c.wt), ptr(c.w), dbl(1e-5))
            else:
                timed("port", O.orc_diff_c, G, o, ptr(c.ut), ptr(c.u), dbl(1e-5)); timed("port", O.orc_diff_c, G, o, ptr(c.vt), ptr(c.v), dbl(1e-5)); timed("port", O.orc_diff_w, G, o, ptr(c.wt), ptr(c.w), dbl(1e-5))
        if cfg["pres"] and with_pres:
            timed("port", O.orc_pres_exec, G, cfg["pres"], ptr(c.p), ptr(pk), ptr(c.u), ptr(c.v), ptr(c.w), ptr(c.ut), ptr(c.vt), ptr(c.wt), ptr(c.rhoref), ptr(c.rhorefh), dbl(1.0))
    step()
    ts = []
    parts = []
    for _ in range(reps):
        tt["ref"] = tt["port"] = 0.0
        t0 = time.perf_counter(); step(); ts.append(time.perf_counter() - t0)
        parts.append((tt["ref"], tt["port"]))
    k = int(np.argsort(ts)[len(ts)//2])
    t = float(ts[k])
    ref_s, port_s = parts[k]
    return {"value": it*jt*kt / t, "unit": "grid-cell updates/s", "cores": 1, "kind": "reference" if use_ref else "port",
            "seconds_per_step": t, "reference_kernels_s": ref_s, "port_s": port_s,
            # the reference's own kernels alone (Advec::exec + Diff::exec + calc_strain2): cells per second of THAT part of the step
            "reference_kernels_rate": (it*jt*kt / ref_s) if (use_ref and ref_s > 0) else None,
            "reference_share_of_step": (ref_s / t) if use_ref else 0.0,
            "sample": ("%s %dx%dx%d (whole grid of this size, same case set-up), %d reps median, 1 thread of %d host cores; "
                       "Advec::exec + Diff::exec + calc_strain2 = the reference's own translation units (oracle/_ref/libmhhref_perf.so, "
                       "-O3 -march=native -DNDEBUG): %.0f %% of the step's time; cyclic fills, N2 + calc_evisc and Pres::exec = the oracle port, same flags "
                       "(pres FFT: the port's radix-2 code, not FFTW): the other %.0f %%" if use_ref else
                       "%s %dx%dx%d, %d reps median, 1 thread of %d host cores; the oracle port only (reference library not present)")
                      % ((case, it, jt, kt, reps, os.cpu_count() or 0, 100.*ref_s/t, 100.*port_s/t) if use_ref else (case, it, jt, kt, reps, os.cpu_count() or 0))}


def recorded_traffic(workload, igc=None):
    """HBM bytes and vector instructions per launch of the dominant kernel from the newest profiles/*_traffic.json that
    scripts/gpu_traffic.sh wrote for this workload -- only if it was measured on the sources this run was built from
    (microhh_amd/stamp.py). Returns (bytes, VALU instructions, source) or (None, None, None)."""
    import glob
    from microhh_amd.stamp import source_stamp
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json"))):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        if d.get("workload") == workload and d.get("stamp") == source_stamp() and d.get("build", "default") == "default" and d.get("igc") == igc:
            best = (f, d)
    if best is None:
        return None, None, None
    return (float(best[1]["total_bytes"]), best[1].get("valu_insts"),
            os.path.relpath(best[0], ROOT) + " (recorded by rocprofv3 PMC passes on these sources, not measured in this run)")


SAMPLER = r"""
import json, re, subprocess, sys, time
for line in sys.stdin:
    if line.strip() != "go":
        continue
    got = []
    for n in range(2):
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showmaxpower"], capture_output=True, text=True, timeout=20).stdout
        except Exception:
            break
        sclk = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", out)
        pw = re.search(r"Current Socket Graphics Package Power \(W\): ([0-9.]+)", out)
        cap = re.search(r"Max Graphics Package Power \(W\): ([0-9.]+)", out)
        if sclk and pw:
            got.append((int(sclk.group(1)), float(pw.group(1)), float(cap.group(1)) if cap else None))
        time.sleep(0.2)
    print(json.dumps(got), flush=True)
"""


def start_power_sampler():
    """A helper process that samples rocm-smi on request, started BEFORE this process touches the GPU (a process that has initialised
    the GPU must not exec another program on this pool; the helper never touches the GPU). None under a profiler (its preloaded library
    initialises the GPU before main() runs) or where rocm-smi is missing."""
    import shutil
    import subprocess
    if not shutil.which("rocm-smi") or "rocprof" in os.environ.get("LD_PRELOAD", "") or os.environ.get("ROCPROFILER_REGISTER_FORCE_LOAD") or os.environ.get("ROCP_TOOL_LIBRARIES"):
        return None
    try:
        return subprocess.Popen([sys.executable, "-c", SAMPLER], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
    except OSError:
        return None


def sample_clock_and_power(sampler, fn, sync, seconds=2.0):
    """Engine clock and socket power while `fn` (one kernel launch) loops: the helper takes two rocm-smi samples while the GPU works
    through a queue of launches. The fused RHS kernel runs at the board's power cap with the engine clock throttled below its 2.4 GHz
    maximum -- the limit its roofs have to be read against (profiles/r3_march_kernel.md)."""
    if sampler is None:
        return None
    for _ in range(30):
        fn()
    sync()
    t0 = time.time()
    try:
        sampler.stdin.write("go\n"); sampler.stdin.flush()
    except OSError:
        return None
    while time.time() - t0 < seconds:                      # keep the queue full while the helper samples
        for _ in range(40):
            fn()
        sync()
    try:
        got = json.loads(sampler.stdout.readline() or "[]")
    except ValueError:
        got = []
    if not got:
        return None
    return {"sclk_mhz": min(g[0] for g in got), "socket_w": max(g[1] for g in got), "cap_w": got[0][2], "samples": len(got),
            "how": "rocm-smi sampled by a helper process while the fused RHS launch loops (outside the timed region)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="drycblles512", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--unfused", action="store_true", help="time Advec::exec + Diff::exec as separate launches")
    ap.add_argument("--device", default="cuda", choices=["cuda", "cpu"],
                    help="cpu = REHEARSAL of this script's control flow (ranks, exchanges, JSON line) with gloo and the test-only CPU "
                         "emulation of the kernels named by MHH_LIB; never a measurement")
    ap.add_argument("--share-gpu", action="store_true",
                    help="REHEARSAL of N > 1 on a one-GPU box: every rank runs its HIP kernels on cuda:0, messages go over gloo "
                         "through host copies; never a measurement")
    ap.add_argument("--build", default="default", choices=["default", "fma"],
                    help="fma = the second, named build (microhh_amd/libmhh_hip_fma.so: marching kernels with FMA contraction; "
                         "tolerance stated in tests/test_fma_build.py) instead of the bit-exact default")
    ap.add_argument("--graph", action="store_true", help="N=1 on a GPU: the timed steps replay one hipGraph of a step (HotPath.capture_step); "
                    "the per-kernel event times then come from untimed call-by-call steps before them")
    ap.add_argument("--no-power-sample", action="store_true", help="skip the rocm-smi clock / power sample (1.6 s of looping the fused RHS launch after the timed region)")
    ap.add_argument("--no-fma-line", action="store_true", help="skip the extra timing of the named FMA build (fma_build in the JSON line)")
    ap.add_argument("--igc", type=int, default=None, help="ghost cells in x (>= the case's own): the layout the reference's grid takes when an operator "
                    "calls Grid::set_minimum_ghost_cells (src/grid.cxx:435-439); 16 = rows of whole 128-byte lines at itot = 512 fp64. A second, NAMED line: "
                    "the headline stays on the reference default")
    ap.add_argument("--force-slab", action="store_true", help="N=1 only: run the slab code path (halo pack/unpack, split pressure solve) with local copies as exchanges")
    args = ap.parse_args()

    if args.build == "fma":
        os.environ["MHH_LIB"] = os.path.join(ROOT, "microhh_amd", "libmhh_hip_fma.so")
    # (before anything touches the GPU)
    sampler = start_power_sampler() if (args.device == "cuda" and int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_power_sample and not args.unfused) else None
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched through torch.distributed.run with %d ranks" % (args.gpus, args.gpus))
    on_gpu = args.device == "cuda"
    # stdout carries exactly ONE line, the JSON: RCCL prints a version banner to stdout when a communicator is created (and
    # gloo its connection notes), so while the job runs file descriptor 1 points at stderr; it is restored for the JSON line.
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)
    if on_gpu:
        if args.share_gpu:
            local = 0
        torch.cuda.set_device(local)
        if world > 1:
            if args.share_gpu: dist.init_process_group("gloo")
            else:              dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        elif os.environ.get("MHH_FORCE_COMM") == "1":
            # one rank, exchanges through RCCL anyway (to self; with --force-slab): what the collectives' call overhead costs per step
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local))
    else:
        if not os.environ.get("MHH_LIB"):
            sys.exit("--device cpu is a rehearsal mode: point MHH_LIB at tests/emul/libmhh_emul.so")
        if world > 1:
            dist.init_process_group("gloo")

    class Ev:                                  # HIP event on the stream the kernels run on; wall clock in the rehearsal
        def __init__(self):
            self.e = torch.cuda.Event(enable_timing=True) if on_gpu else None
            self.t = 0.0

        def record(self):
            if on_gpu:
                self.e.record()
            else:
                self.t = time.perf_counter()

        def elapsed_time(self, other):
            return self.e.elapsed_time(other.e) if on_gpu else 1e3 * (other.t - self.t)

    case, itot, jtot, ktot, desc = WORKLOADS[args.workload]
    from microhh_amd.model import HotPath
    rhs_only = args.workload in FP32_RHS_ONLY
    hp = HotPath(case, itot, jtot, ktot, device=("cuda:%d" % local if on_gpu else "cpu"), npy=world, rank=rank,   # slab in y: npx=1, npy=world
                 force_slab=(args.force_slab and world == 1), dtype=(np.float32 if rhs_only else np.float64), igc=args.igc)

    rhs = hp.rhs_unfused if args.unfused else hp.rhs

    overlapped = hp.can_overlap and not args.unfused

    def one_step(ev=None):
        if overlapped:                   # N > 1: the prognostic halo exchange travels behind the interior rows
            hp.halo_visc_rhs(ev)
            if not rhs_only:
                hp.pres()
            return
        hp.cyclic_prognostic()
        if ev is not None:
            ev[2].record()
        hp.exec_viscosity()
        if ev is not None:
            ev[0].record()
        rhs()
        if ev is not None:
            ev[1].record()
        if not rhs_only:
            hp.pres()
            if ev is not None:
                ev[3].record()

    for _ in range(args.warmup):
        one_step()
    events = [tuple(Ev() for _ in range(4)) for _ in range(args.steps)]

    def barrier():
        if on_gpu:
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            if on_gpu:
                torch.cuda.synchronize()
    barrier()
    if on_gpu and (world > 1 or os.environ.get("MHH_FORCE_COMM") == "1") and not args.share_gpu:
        hp.comm_timing = []                  # device-event pairs around every exchange of the timed steps
    use_graph = args.graph and on_gpu and world == 1 and not hp.slab and not args.unfused
    if use_graph:
        for n in range(args.steps):          # event times of the kernels: call-by-call steps, outside the timed region
            one_step(events[n])
        graph = hp.capture_step()
        barrier()
    t0 = time.perf_counter()
    for n in range(args.steps):
        if use_graph:
            graph.replay()
        else:
            one_step(events[n])
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=("cuda" if on_gpu and not args.share_gpu else "cpu"), dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms = 1e3 * elapsed / args.steps
    cells = itot * jtot * ktot
    rhs_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in events]))   # fused RHS kernel, this rank, ms per launch
    visc_ms = float(np.mean([e[2].elapsed_time(e[0]) for e in events]))  # exec_viscosity (+ its cyclic fill), ms per call
    local_cells = hp.grid.imax * hp.grid.jmax * hp.grid.kmax
    if overlapped:                       # the timed launch covers the interior rows only
        local_cells_rhs = hp.grid.imax * hp.rhs_rows_timed * hp.grid.kmax
    else:
        local_cells_rhs = local_cells
    alg_bytes = hp.alg_bytes_rhs() * local_cells_rhs
    achieved = alg_bytes / (rhs_ms * 1e-3) / 1e9
    out = {
        "metric": "grid-cell updates/sec (full RHS+pres step)" if not rhs_only else "grid-cell updates/sec (RHS: exec_viscosity + advec + diff)", "value": cells / (elapsed / args.steps), "unit": "grid-cell updates/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32" if rhs_only else "f64", "data": "synthetic",
        "config": {"workload": desc, "grid": [itot, jtot, ktot], "decomposition": "slab-y npx=1 npy=%d" % world,
                   "ghost_cells": [hp.grid.igc, hp.grid.jgc, hp.grid.kgc], "rhs": "unfused" if args.unfused else "fused",
                   **({"pres_k_slices": hp.pres_chunks, "pres_form": "x and y transforms in LDS on the transposes' buffers" if (hp.slim and hp.lib.mhh_pres_slab_has_lds(hp.plan) == 1) else "staged (rocFFT)",
                       "note": "halo overlap and k-sliced transposes are the defaults for N > 1 without a multi-GPU measurement behind them (no node was available to the builder); "
                               "MHH_OVERLAP=0 / MHH_PRES_CHUNKS=1 switch them off"} if hp.slab else {}), "halo_overlap": bool(overlapped), "launch": "hipGraph replay" if use_graph else "call by call"},
        "roofline": {"bound": "hbm", "kernel": "fused RHS (advec+diff) pass" if not args.unfused else "advec+diff launches",
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": None, "alg_bytes_per_cell": hp.alg_bytes_rhs(), "ms_per_launch": rhs_ms},
        # the pair BASELINE.json's target is quoted on: exec_viscosity (5s B/cell) + fused tendencies (13s B/cell) = 18s
        "rhs_with_viscosity": {"alg_bytes_per_cell": hp.alg_bytes_rhs() + hp.alg_bytes_visc(), "ms": rhs_ms + visc_ms,
                               "frac": (hp.alg_bytes_rhs() + hp.alg_bytes_visc()) * local_cells_rhs / ((rhs_ms + visc_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS},
        "alg_bytes_per_cell_full_step": hp.alg_bytes_rhs() + hp.alg_bytes_visc() + (0 if rhs_only else hp.alg_bytes_pres()),
        "hbm_frac_full_step": (hp.alg_bytes_rhs() + hp.alg_bytes_visc() + (0 if rhs_only else hp.alg_bytes_pres())) * local_cells / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
    }
    if not rhs_only:
        # self-check of the solve on the state the timed steps left behind (before anything else launches kernels on these fields) (every rank takes part: halos + MAX over ranks):
        # Pres::input -- the divergence of u/dt + ut, which Pres::exec has just removed -- beside the same of u, v, w alone
        comm_events, hp.comm_timing = hp.comm_timing, None        # (the exchanges of the check are not the step's)
        d1, d0 = hp.projected_divergence()
        hp.comm_timing = comm_events
        out["self_check"] = {"max_abs_pres_input_after_pres": d1, "max_abs_pres_input_of_u_over_dt": d0, "ratio": d1 / d0 if d0 else None}
    if not rhs_only and not overlapped and not hp.slab:
        out["pressure"] = {"ms": float(np.mean([e[1].elapsed_time(e[3]) for e in events])),
                           "form": "transforms in LDS, 3 kernels (csrc/pres_lds.h, pres_lds4.h)" if hp.lib.mhh_pres_exec_form(hp.plan) == 1 else "staged, rocFFT (csrc/k_pres.hip)"}
    out["build"] = args.build if not os.environ.get("MHH_LIB") or args.build == "fma" else os.path.basename(os.environ["MHH_LIB"])
    valu_insts = None
    if world == 1 and not args.unfused and out["build"] == "default":
        out["roofline"]["traffic"], valu_insts, src = recorded_traffic(args.workload, args.igc)      # FETCH_SIZE x 2 + WRITE_SIZE per launch, or null
        if src:
            out["roofline"]["traffic_source"] = src
    # The second roof of the dominant kernel: vector-instruction issue. A wave64 fp64 instruction occupies its SIMD for 4 cycles;
    # 1024 SIMDs. `insts` = SQ_INSTS_VALU per launch from the same stamped PMC record as `traffic`; the floor is given at the chip's
    # maximum engine clock and at the clock it actually holds under this kernel (sampled live below: the board runs it at its power cap).
    power = None
    if world == 1 and on_gpu and not args.unfused and not args.no_power_sample:
        power = sample_clock_and_power(sampler, rhs, lambda: torch.cuda.synchronize())
        if power:
            out["power"] = power
    if valu_insts:
        f_max = 2.4e9
        f_now = (power["sclk_mhz"] * 1e6) if power else None
        floor = lambda f: valu_insts * 4.0 / (1024 * f) * 1e3           # ms  # noqa: E731
        out["roofline"]["valu"] = {"bound": "vector-instruction issue (wave64 fp64: 4 cycles per instruction and SIMD, 1024 SIMDs)",
                                   "insts_per_launch": valu_insts, "insts_per_cell": valu_insts * 64.0 / local_cells_rhs,
                                   "floor_ms_at_2400_mhz": floor(f_max), "frac_at_2400_mhz": floor(f_max) / rhs_ms,
                                   "floor_ms_at_sampled_clock": floor(f_now) if f_now else None,
                                   "frac_at_sampled_clock": (floor(f_now) / rhs_ms) if f_now else None}
    if world == 1 and on_gpu and args.build == "default" and not args.unfused and not rhs_only and not args.no_fma_line and not os.environ.get("MHH_LIB") \
            and os.path.exists(os.path.join(ROOT, "microhh_amd", "libmhh_hip_fma.so")):
        # the same fused RHS launch from the named FMA build, timed the same way (its own fields; nothing of it enters `value`)
        from microhh_amd import capi
        fl = capi.bind(C.CDLL(os.path.join(ROOT, "microhh_amd", "libmhh_hip_fma.so")))
        hp2 = HotPath(case, itot, jtot, ktot, dtype=np.float64, device="cuda:%d" % local, lib=fl)
        hp2.cyclic_prognostic(); hp2.exec_viscosity()
        for _ in range(3):
            hp2.rhs()
        ev2 = [(Ev(), Ev()) for _ in range(min(args.steps, 20))]
        for a_, b_ in ev2:
            a_.record(); hp2.rhs(); b_.record()
        torch.cuda.synchronize()
        fms = float(np.mean([a_.elapsed_time(b_) for a_, b_ in ev2]))
        out["fma_build"] = {"library": "microhh_amd/libmhh_hip_fma.so", "kernel": "fused RHS (advec+diff) pass", "ms_per_launch": fms,
                            "frac": alg_bytes / (fms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            "tolerance": "tendency increments within 64 ulp of the field's largest increment of the bit-exact build (tests/test_fma_build.py)"}
        hp2.close()
    if not on_gpu:
        out["data"] = "synthetic; CPU REHEARSAL of the script (emulated kernels, gloo): not a measurement"
    if hp.comm_timing:
        # where the time of an N > 1 step goes on this rank: stream time inside the exchanges (the launch of the collective until
        # its result is usable by the next kernel), per step; the rest of ms_per_step is kernels
        tags = sorted({t for t, _, _ in hp.comm_timing})
        out["comm"] = {t + "_ms_per_step": sum(a.elapsed_time(b) for tt, a, b in hp.comm_timing if tt == t) / args.steps for t in tags}
        out["comm"]["exchanges_per_step"] = len(hp.comm_timing) / args.steps
    if on_gpu and args.share_gpu:
        out["data"] = "synthetic; REHEARSAL of the N > 1 path with all ranks on one GPU (gloo, host-staged messages): not a measurement"
    if rank == 0 and world == 1 and not args.no_cpu_baseline and on_gpu:
        out["cpu_baseline"] = cpu_baseline("drycblles" if case == "gabls1" else case, sample=(min(itot, 256), min(jtot, 256), min(ktot, 256)), with_pres=not rhs_only)
    if sampler is not None:
        try:
            sampler.stdin.close(); sampler.wait(timeout=10)
        except Exception:
            pass
    hp.close()
    if dist.is_initialized():
        dist.destroy_process_group()
    sys.stdout.flush()
    os.dup2(stdout_fd, 1); os.close(stdout_fd)
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
