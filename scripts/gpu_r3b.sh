#!/bin/bash
# scripts/gpu_r3b.sh -- round 3, call b: the marching kernel after the copy-section clean-up (padded tile slots, scalar sweep
# predicate, power-of-two ring slot, no plane step-back in interior levels): parity, bench, phase stamps.
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r3b
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_parity.py tests/test_golden.py -m gpu -q -x > $OUT/pytest.log 2>&1; rc=$?
tail -3 $OUT/pytest.log
[ $rc -ne 0 ] && { echo "pytest failed: stopping"; exit $rc; }
run() {
  local name=$1 lib=$2; shift 2
  MHH_LIB=${lib:+$PWD/$lib} timeout -k 10 300 python bench.py --workload drycblles512 --steps 20 --warmup 3 --no-cpu-baseline "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err || { echo "bench $name failed"; tail -3 $OUT/bench_$name.err; return 1; }
  python - "$OUT/bench_$name.json" "$name" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print("%-28s ms/step %7.3f  rhs ms %7.3f  rhs-frac %.3f  fma %s" % (sys.argv[2], d["ms_per_step"], d["roofline"]["ms_per_launch"], d["roofline"]["frac"], d.get("fma_build",{}).get("ms_per_launch")))
PY
}
run default_igc3 "" && run default_igc16 "" --igc 16 || exit 3
for v in $(ls microhh_amd/variants/*.so | grep -v stamp); do run $(basename $v .so) $v --no-fma-line || exit 3; done
echo "== stamps igc 3"; MHH_LIB=$PWD/microhh_amd/variants/libmhh_hip_stamp.so timeout -k 10 300 python scripts/experiments/march_stamps.py > $OUT/stamps_igc3.txt 2>&1; cat $OUT/stamps_igc3.txt
echo "== stamps igc 16"; MHH_LIB=$PWD/microhh_amd/variants/libmhh_hip_stamp.so timeout -k 10 300 python scripts/experiments/march_stamps.py --igc 16 > $OUT/stamps_igc16.txt 2>&1; cat $OUT/stamps_igc16.txt
