"""Source stamp of the HIP library: a hash over the kernel sources and the C-ABI header. Profiles that carry it
(profiles/*_traffic.json, written by scripts/gpu_traffic.sh) can be matched against the sources a bench run was built from."""
import hashlib
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def source_stamp():
    h = hashlib.sha256()
    csrc = os.path.join(HERE, "csrc")
    files = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".h")))
    files.append(os.path.join(HERE, "..", "include", "mhh_hip.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(source_stamp())
