#!/bin/bash
# scripts/gpu_traffic.sh -- HBM traffic (FETCH_SIZE x 2 + WRITE_SIZE, two PMC passes) and duration of the kernels matching a
# pattern, for the default library and every build under microhh_amd/variants/.
set -o pipefail
# BENCH_ARGS: extra bench.py arguments (e.g. "--igc 16"); ONLY_DEFAULT=1: the default library only. 12 launches per pass.
TAG=${1:-traffic}; WL=${2:-drycblles512}; PAT=${3:-rhs25_march}
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
VARIANTS=$(ls microhh_amd/variants/*.so 2>/dev/null); [ -n "$ONLY_DEFAULT" ] && VARIANTS=""
for v in "" $VARIANTS; do
  name=$(basename "${v:-default}" .so)
  export MHH_LIB=${v:+$PWD/$v}
  for c in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
    timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$name/$c -- python3 bench.py --workload $WL --steps 10 --warmup 2 --no-cpu-baseline --no-fma-line --no-power-sample $BENCH_ARGS > $OUT/$name.$c.json 2> $OUT/$name.$c.err || { echo "$name $c failed"; tail -2 $OUT/$name.$c.err; }
  done
  python3 - $OUT/$name "$PAT" "$name" "$WL" "$BENCH_ARGS" <<'PY'
import csv, glob, sys
tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"):
    vals = []
    for f in glob.glob(sys.argv[1] + "/" + c + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if sys.argv[2] in r["Kernel_Name"] and r["Counter_Name"] == c: vals.append(float(r["Counter_Value"]))
    tot[c] = sum(vals)/max(1, len(vals))
print("%-24s fetch %.2f GB  write %.2f GB  total %.2f GB" % (sys.argv[3], tot["FETCH_SIZE"]*2*1024/1e9, tot["WRITE_SIZE"]*1024/1e9, (tot["FETCH_SIZE"]*2 + tot["WRITE_SIZE"])*1024/1e9))
# the record bench.py reads back (copy it to profiles/<round>_<workload>_traffic.json): bytes per launch + the source stamp
import json, os
sys.path.insert(0, os.getcwd())
from microhh_amd.stamp import source_stamp
import re
m = re.search(r"--igc (\d+)", sys.argv[5] if len(sys.argv) > 5 else "")
json.dump({"valu_insts": tot["SQ_INSTS_VALU"], "workload": sys.argv[4], "kernel": sys.argv[2], "build": sys.argv[3], "stamp": source_stamp(), "igc": int(m.group(1)) if m else None, "launches_averaged": len(vals),
           "fetch_bytes": tot["FETCH_SIZE"]*2*1024, "write_bytes": tot["WRITE_SIZE"]*1024, "total_bytes": (tot["FETCH_SIZE"]*2 + tot["WRITE_SIZE"])*1024,
           "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes with --kernel-trace, averages per launch; FETCH_SIZE (KB) doubled for gfx950 (MI355X_MICROARCH.md, HBM); Infinity-Cache hits are counted, not excluded"},
          open(sys.argv[1] + "_traffic.json", "w"), indent=1)
PY
done
