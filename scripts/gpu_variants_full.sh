#!/bin/bash
# scripts/gpu_variants_full.sh -- like gpu_variants.sh, also printing the exec_viscosity + RHS pair time (tiling sweeps).
set -o pipefail
TAG=${1:-variants}; WL=${2:-drycblles512}
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
for v in "" $(ls microhh_amd/variants/*.so 2>/dev/null); do
  name=$(basename "${v:-default}" .so)
  MHH_LIB=${v:+$PWD/$v} timeout -k 10 300 python bench.py --workload $WL --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_${WL}_$name.json 2> $OUT/bench_$name.err || { echo "bench $name failed"; tail -3 $OUT/bench_$name.err; continue; }
  python - "$OUT/bench_${WL}_$name.json" "$name" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print("%-28s ms/step %7.3f  rhs ms %7.3f  visc+rhs ms %7.3f" % (sys.argv[2], d["ms_per_step"], d["roofline"]["ms_per_launch"], d["rhs_with_viscosity"]["ms"]))
PY
done
