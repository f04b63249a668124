#!/usr/bin/env python3
"""scripts/isa_loop_stats.py <disassembly.s> <kernel-name-substring> -- instruction mix of a kernel's loops from
`llvm-objdump -d` output of the gfx950 code object: for every backward branch (a loop), the instructions between its target
and the branch, counted by class (VALU fp64 arithmetic, other VALU, v_readlane/v_writelane, SALU, s_load, ds_*, global_*,
scratch_*, s_waitcnt, s_barrier). The biggest loop of rhs25_march_kernel is the rotated six-level group.
    /opt/rocm/lib/llvm/bin/llvm-objdump --offloading k_march.o ; llvm-objdump -d k_march.o.0.hipv4-* > k.s"""
import re
import sys
from collections import Counter


def classify(op):
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")): return "v_lane"
    if op.startswith("v_") and op.endswith("_f64") or op.startswith(("v_fma_f64", "v_mul_f64", "v_add_f64")): return "valu_f64"
    if op.startswith("v_pk_"): return "valu_pk"
    if op.startswith("v_"): return "valu_other"
    if op.startswith("s_waitcnt"): return "s_waitcnt"
    if op.startswith("s_barrier"): return "s_barrier"
    if op.startswith("s_load") or op.startswith("s_buffer_load"): return "s_load"
    if op.startswith(("s_cbranch", "s_branch")): return "s_branch"
    if op.startswith("s_nop"): return "s_nop"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith("global_load_lds"): return "lds_dma"
    if op.startswith("global_load"): return "global_load"
    if op.startswith("global_store"): return "global_store"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith("buffer_"): return "buffer"
    return "other"


def main():
    path, pat = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    # functions: "<addr> <name>:" headers
    hdr = re.compile(r"^([0-9a-f]+) <(.+)>:$")
    ins = re.compile(r"^\s+(\S+)\s.*//\s*([0-9A-Fa-f]+):")
    cur, funcs = None, {}
    for ln in lines:
        m = hdr.match(ln)
        if m:
            cur = m.group(2); funcs[cur] = []
            continue
        if cur is None:
            continue
        m = ins.match(ln)
        if m:
            funcs[cur].append((int(m.group(2), 16), m.group(1), ln))
    for name, body in funcs.items():
        if pat not in name or not body:
            continue
        print("== %s: %d instructions" % (name[:110], len(body)))
        tot = Counter(classify(op) for _, op, _ in body)
        print("   whole kernel:", dict(tot))
        addr_index = {a: n for n, (a, _, _) in enumerate(body)}
        loops = []
        for n, (a, op, ln) in enumerate(body):
            if op.startswith("s_cbranch") or op == "s_branch":
                m = re.search(r"<[^>]*\+0x([0-9a-f]+)>", ln)
                if not m:
                    continue
                tgt = body[0][0] + int(m.group(1), 16)
                # objdump prints offsets relative to the symbol; fall back on absolute match
                if tgt not in addr_index:
                    continue
                t = addr_index[tgt]
                if t < n:
                    loops.append((n - t + 1, t, n))
        # barrier-to-barrier segments: in the marching kernels one segment = one level (the reliable per-level mix: the loop
        # ranges below also contain pre-headers and exit paths that sit inside the loop's address range)
        bars = [n for n, (_, op, _) in enumerate(body) if op.startswith("s_barrier")]
        for a_, b_ in zip(bars[:-1], bars[1:]):
            c = Counter(classify(op) for _, op, _ in body[a_:b_])
            valu = c["valu_f64"] + c["valu_other"] + c["valu_pk"] + c["v_lane"]
            print("   barrier @%x -> next: %5d instr, VALU %4d (f64 %4d, other %3d, lane %3d)  SALU %3d s_load %2d waitcnt %2d nop %2d branch %2d  LDS %2d DMA %2d gload %2d gstore %2d scratch %d"
                  % (body[a_][0], b_ - a_, valu, c["valu_f64"], c["valu_other"] + c["valu_pk"], c["v_lane"], c["salu"], c["s_load"], c["s_waitcnt"], c["s_nop"], c["s_branch"],
                     c["lds"], c["lds_dma"], c["global_load"], c["global_store"], c["scratch"]))
        loops.sort(reverse=True)
        for size, t, n in loops[:6]:
            c = Counter(classify(op) for _, op, _ in body[t:n+1])
            print("   loop of %5d instructions @%x: %s" % (size, body[t][0], dict(c)))


if __name__ == "__main__":
    main()
