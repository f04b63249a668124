#!/bin/bash
# scripts/gpu_kernel_times.sh -- per-kernel average durations (rocprofv3 --kernel-trace --stats) of one bench workload,
# for the default library and every build under microhh_amd/variants/.
set -o pipefail
TAG=${1:-kt}; WL=${2:-drycblles512}; FILTER=${3:-.}
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
for v in "" $(ls microhh_amd/variants/*.so 2>/dev/null); do
  name=$(basename "${v:-default}" .so)
  export MHH_LIB=${v:+$PWD/$v}
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -- python3 bench.py --workload $WL --steps 5 --warmup 2 --no-cpu-baseline > $OUT/$name.json 2> $OUT/$name.err || { echo "$name failed"; tail -3 $OUT/$name.err; continue; }
  echo "== $name"
  python3 - $OUT/$name "$FILTER" <<'PY'
import csv, glob, sys, re
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if re.search(sys.argv[2], r["Name"]) and float(r["AverageNs"]) > 2e4 and "at::native" not in r["Name"]:
            print("  %9.1f us x%-3s %s" % (float(r["AverageNs"])/1e3, r["Calls"], r["Name"][:90]))
PY
done
