#!/bin/bash
# scripts/gpu_ab3.sh [tag] [workload] [rounds] -- careful A/B on ONE box: the default library and every build under
# microhh_amd/variants/, round-robin, `rounds` times (default 3); prints min / median fused-RHS ms per build. The chip runs this
# kernel at its power cap, so single runs scatter by ~1 %; box-to-box by 3-6 %.
set -o pipefail
export TMPDIR=/tmp
TAG=${1:-ab3}; WL=${2:-drycblles512}; R=${3:-3}
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_parity.py -m gpu -q -x -k "fused_rhs or marching or sixteen" > $OUT/pytest.log 2>&1; rc=$?
tail -1 $OUT/pytest.log
[ $rc -ne 0 ] && { echo "pytest failed: stopping"; exit $rc; }
NAMES="default $(ls microhh_amd/variants/*.so 2>/dev/null | xargs -n1 basename 2>/dev/null | sed 's/libmhh_hip_//; s/\.so//')"
for r in $(seq 1 $R); do
  for name in $NAMES; do
    if [ "$name" = default ]; then unset MHH_LIB; else export MHH_LIB=$PWD/microhh_amd/variants/libmhh_hip_$name.so; fi
    timeout -k 10 300 python bench.py --workload $WL --steps 30 --warmup 5 --no-cpu-baseline --no-fma-line $BENCH_ARGS > $OUT/bench_${name}_$r.json 2> $OUT/bench_${name}_$r.err || { echo "bench $name failed"; tail -3 $OUT/bench_${name}_$r.err; exit 3; }
  done
done
unset MHH_LIB
python - $OUT $R $NAMES <<'PY'
import json, sys, statistics
out, R, names = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
for n in names:
    v = [json.load(open("%s/bench_%s_%d.json" % (out, n, r)))["roofline"]["ms_per_launch"] for r in range(1, R+1)]
    s = [json.load(open("%s/bench_%s_%d.json" % (out, n, r)))["ms_per_step"] for r in range(1, R+1)]
    print("%-16s rhs ms min %.3f median %.3f  (%s)   step median %.3f" % (n, min(v), statistics.median(v), " ".join("%.3f" % x for x in v), statistics.median(s)))
PY
