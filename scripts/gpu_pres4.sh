#!/bin/bash
# scripts/gpu_pres4.sh [tag] -- the pres_4 LDS form on the GPU: parity tests, then moser600 with the staged and the LDS form, round-robin
set -o pipefail
export TMPDIR=/tmp
TAG=${1:-pres4}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_parity.py tests/test_taylorgreen.py -m gpu -q -x -k "pres or taylor" > $OUT/pytest.log 2>&1; rc=$?
tail -3 $OUT/pytest.log
[ $rc -ne 0 ] && { echo "pytest failed: stopping"; tail -40 $OUT/pytest.log; exit $rc; }
for r in 1 2 3; do
  for form in 0 1; do
    MHH_PRES_LDS=$form timeout -k 10 300 python bench.py --workload moser600 --steps 30 --warmup 5 --no-cpu-baseline --no-fma-line --no-power-sample > $OUT/bench_lds${form}_$r.json 2> $OUT/bench_lds${form}_$r.err || { echo "bench failed"; tail -5 $OUT/bench_lds${form}_$r.err; exit 3; }
    python - $OUT/bench_lds${form}_$r.json $form <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("MHH_PRES_LDS=%s: step %.3f ms, pressure %.3f ms (%s), self_check %s" % (sys.argv[2], d["ms_per_step"], d["pressure"]["ms"], d["pressure"]["form"], d.get("self_check")))
PY
  done
done
