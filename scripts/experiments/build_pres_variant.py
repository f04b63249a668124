"""Variant builds of k_pres.hip alone, linked with the objects of the default build: microhh_amd/variants/libmhh_hip_<tag>.so.
python scripts/experiments/build_pres_variant.py tag -DFLAG=.. [...]"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from microhh_amd import build as b
tag, flags = sys.argv[1], sys.argv[2:]
vdir = os.path.join(b.HERE, "variants"); os.makedirs(vdir, exist_ok=True)
obj = os.path.join(vdir, "k_pres_%s.o" % tag)
subprocess.run([b.HIPCC] + b.CFLAGS + flags + ["-c", os.path.join(b.CSRC, "k_pres.hip"), "-o", obj], check=True)
others = [os.path.join(b.CSRC, s.replace(".hip", ".o")) for s in b.SOURCES if s != "k_pres.hip"]
lib = os.path.join(vdir, "libmhh_hip_%s.so" % tag)
subprocess.run([b.HIPCC, "--offload-arch=gfx950", "-shared", "-o", lib, obj] + others + ["-L/opt/rocm/lib", "-lrocfft", "-Wl,-rpath,/opt/rocm/lib"], check=True)
os.remove(obj)
print(lib)
