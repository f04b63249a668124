#!/bin/bash
# scripts/gpu_ab.sh -- parity tests + A/B of kernel implementations / variant builds on the GPU box.
set -o pipefail
TAG=${1:-ab}; shift
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
echo "== pytest -m gpu"; timeout -k 10 900 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; rc=$?
tail -4 $OUT/pytest_gpu.log
if [ $rc -gt 1 ]; then echo "pytest crashed (rc=$rc): stopping"; exit $rc; fi
run() { # name, env..., -- bench args
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err || { echo "bench $name failed"; tail -3 $OUT/bench_$name.err; return; }
  python - "$OUT/bench_$name.json" "$name" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print("%-28s ms/step %7.3f  rhs ms %7.3f  value %.3e  rhs-frac %.3f" % (sys.argv[2], d["ms_per_step"], d["roofline"]["ms_per_launch"], d["value"], d["roofline"]["frac"]))
PY
}
run march512 X=1 -- --workload drycblles512
run march512_nodma MHH_MARCH_DMA=0 -- --workload drycblles512
run cell512 MHH_RHS25_IMPL=cell -- --workload drycblles512
run march256 X=1 -- --workload drycblles256
run cell256 MHH_RHS25_IMPL=cell -- --workload drycblles256
for v in $(ls microhh_amd/variants/*.so 2>/dev/null); do
  run $(basename $v .so)_512 MHH_LIB=$PWD/$v -- --workload drycblles512
done
