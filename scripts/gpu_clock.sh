#!/bin/bash
# scripts/gpu_clock.sh -- effective clock (GRBM_GUI_ACTIVE / 8 / duration) and instruction-cache counters of the march kernel,
# default library and every variant build.
set -o pipefail
TAG=${1:-clock}; WL=${2:-drycblles512}; PAT=${3:-rhs25_march}
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
for v in "" $(ls microhh_amd/variants/*.so 2>/dev/null); do
  name=$(basename "${v:-default}" .so)
  export MHH_LIB=${v:+$PWD/$v}
  timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_IFETCH --kernel-trace --output-format csv -d $OUT/$name -- python3 bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-fma-line > $OUT/$name.json 2> $OUT/$name.err || { echo "$name failed"; tail -2 $OUT/$name.err; }
  python3 - $OUT/$name "$PAT" "$name" <<'PY'
import csv, glob, sys, collections
vals = collections.defaultdict(list); dur = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
m = {k: sum(v)/len(v) for k, v in vals.items()}
d = sum(dur)/max(1, len(dur))
cyc = m.get("GRBM_GUI_ACTIVE", 0)/8
print("%-22s %.3f ms  %.2f Mcycles  %.2f GHz  icache req %.3g miss %.3g  wait %.2f  valu busy %.2f  ifetch %.3g" % (
    sys.argv[3], d, cyc/1e6, cyc/(d*1e6) if d else 0, m.get("SQC_ICACHE_REQ", 0), m.get("SQC_ICACHE_MISSES", 0),
    m.get("SQ_WAIT_ANY", 0)/max(1, m.get("SQ_WAVE_CYCLES", 1)), m.get("SQ_ACTIVE_INST_VALU", 0)*4/(1024*max(cyc, 1)), m.get("SQ_IFETCH", 0)))
PY
done
