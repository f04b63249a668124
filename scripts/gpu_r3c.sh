#!/bin/bash
# scripts/gpu_r3c.sh [tag] -- A/B of every build under microhh_amd/variants/ against the default library on ONE box (box-to-box
# spread is +-3 %): fused-RHS ms from bench.py (20 steps), then the phase stamps of the stamp variant if present.
set -o pipefail
export TMPDIR=/tmp
TAG=${1:-r3c}; WL=${2:-drycblles512}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_parity.py -m gpu -q -x -k "fused_rhs or marching or sixteen" > $OUT/pytest.log 2>&1; rc=$?
tail -2 $OUT/pytest.log
[ $rc -ne 0 ] && { echo "pytest failed: stopping"; exit $rc; }
run() {
  local name=$1 lib=$2; shift 2
  MHH_LIB=${lib:+$PWD/$lib} timeout -k 10 300 python bench.py --workload $WL --steps 20 --warmup 3 --no-cpu-baseline --no-fma-line "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err || { echo "bench $name failed"; tail -3 $OUT/bench_$name.err; return 1; }
  python - "$OUT/bench_$name.json" "$name" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print("%-28s ms/step %7.3f  rhs ms %7.3f  rhs-frac %.3f" % (sys.argv[2], d["ms_per_step"], d["roofline"]["ms_per_launch"], d["roofline"]["frac"]))
PY
}
run default "" || exit 3
for v in $(ls microhh_amd/variants/*.so 2>/dev/null | grep -v stamp); do run $(basename $v .so | sed s/libmhh_hip_//) $v || exit 3; done
run default_again "" || exit 3
if [ -f microhh_amd/variants/libmhh_hip_stamp.so ]; then
  echo "== stamps"; MHH_LIB=$PWD/microhh_amd/variants/libmhh_hip_stamp.so timeout -k 10 300 python scripts/experiments/march_stamps.py > $OUT/stamps.txt 2>&1; grep -v amdgpu.ids $OUT/stamps.txt
fi
