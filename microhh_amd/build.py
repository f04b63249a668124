"""Build libmhh_hip.so: hand-written HIP kernels + C-ABI, cross-compiled for gfx950 with hipcc.

    python -m microhh_amd.build            # strict build (default): -ffp-contract=off, bit-exact stencils
The .so is written in-tree (microhh_amd/libmhh_hip.so) so that it travels with the repo snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["k_stencil.hip", "k_rhs.hip", "k_march.hip", "k_march4.hip", "k_visc.hip", "k_pres.hip", "k_slab.hip"]
LIB = os.path.join(HERE, "libmhh_hip.so")
# Second, NAMED build: the marching kernels with FMA contraction allowed (-ffp-contract=fast), everything else the same objects.
# Not bit-identical to the pinned rounding of the oracle: tests/test_fma_build.py states and checks its tolerance. Selected with
# MHH_LIB=<this file> (bench.py --build fma); the default library stays the bit-exact one.
LIB_FMA = os.path.join(HERE, "libmhh_hip_fma.so")
FMA_SOURCES = ["k_march.hip", "k_march4.hip", "k_visc.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: no FMA contraction, so that every stencil rounds like the reference CPU path built
# with the same setting (DESIGN.md "Parity").
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden",
          "-Wall", "-Wno-unused-function", "-I/opt/rocm/include", "-I" + CSRC,
          # gfx950_prims.h declares M0 clobbered by the LDS-DMA statements; the backend notes per statement that M0 is a reserved
          # register (5568 notes per build) -- the declaration is what we want (an implicit def of M0 the compiler's own M0 users see)
          "-Wno-inline-asm"]


def _newer(src, dst):
    return (not os.path.exists(dst)) or os.path.getmtime(src) > os.path.getmtime(dst)


def build_variant(tag, extra_flags):
    """Experiment builds (tuning sweeps): microhh_amd/variants/libmhh_hip_<tag>.so, never loaded by default."""
    vdir = os.path.join(HERE, "variants")
    os.makedirs(vdir, exist_ok=True)
    lib = os.path.join(vdir, "libmhh_hip_%s.so" % tag)
    flags = [f for f in CFLAGS if not (f.startswith("-ffp-contract") and any(e.startswith("-ffp-contract") for e in extra_flags))]
    cmd = [HIPCC] + flags + list(extra_flags) + ["-shared", "-o", lib] + [os.path.join(CSRC, s) for s in SOURCES] + ["-L/opt/rocm/lib", "-lrocfft", "-Wl,-rpath,/opt/rocm/lib"]
    print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return lib


def build_variant_of(tag, extra_flags, sources=("k_march.hip",)):
    """Experiment build that recompiles only `sources` with `extra_flags` and links them with the default build's other objects
    (run build() first): microhh_amd/variants/libmhh_hip_<tag>.so. Minutes instead of a quarter of an hour per variant."""
    vdir = os.path.join(HERE, "variants")
    os.makedirs(vdir, exist_ok=True)
    lib = os.path.join(vdir, "libmhh_hip_%s.so" % tag)
    flags = [f for f in CFLAGS if not (f.startswith("-ffp-contract") and any(e.startswith("-ffp-contract") for e in extra_flags))]
    objs = []
    for s in SOURCES:
        if s in sources:
            obj = os.path.join(vdir, s.replace(".hip", ".%s.o" % tag))
            cmd = [HIPCC] + flags + list(extra_flags) + ["-c", os.path.join(CSRC, s), "-o", obj]
            print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        else:
            obj = os.path.join(CSRC, s.replace(".hip", ".o"))
        objs.append(obj)
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-L/opt/rocm/lib", "-lrocfft", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return lib


def build(force=False, verbose=True):
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(HERE, "..", "include", "mhh_hip.h")]
    objs, jobs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _newer(src, obj) or any(_newer(d, obj) for d in deps):
            jobs.append([HIPCC] + CFLAGS + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    fma_objs, fma_jobs = [], []
    fflags = [f for f in CFLAGS if not f.startswith("-ffp-contract")] + ["-ffp-contract=fast", "-DMHH_FMA_BUILD=1"]
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        if s in FMA_SOURCES:
            obj = os.path.join(CSRC, s.replace(".hip", ".fma.o"))
            if force or _newer(src, obj) or any(_newer(d, obj) for d in deps):
                fma_jobs.append([HIPCC] + fflags + ["-c", src, "-o", obj])
        else:
            obj = os.path.join(CSRC, s.replace(".hip", ".o"))
        fma_objs.append(obj)
    with ThreadPoolExecutor(max_workers=6) as ex:
        list(ex.map(run, jobs + fma_jobs))
    link = ["-L/opt/rocm/lib", "-lrocfft", "-Wl,-rpath,/opt/rocm/lib"]
    if jobs or not os.path.exists(LIB):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + link)
    if jobs or fma_jobs or not os.path.exists(LIB_FMA):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_FMA] + fma_objs + link)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
