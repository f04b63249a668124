#!/bin/bash
# scripts/gpu_check.sh -- what one gpurun call does on the GPU box: parity tests, smoke, bench lines, rocprof stats.
# Usage (from the repo root on the box): bash scripts/gpu_check.sh [tag]
set -o pipefail
TAG=${1:-r1}
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
echo "== pytest -m gpu"; timeout -k 10 900 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; rc=$?
tail -15 $OUT/pytest_gpu.log
if [ $rc -gt 1 ]; then echo "pytest crashed (rc=$rc): stopping"; exit $rc; fi
echo "== smoke"; timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 || { tail -20 $OUT/smoke.log; echo "smoke failed"; }
tail -2 $OUT/smoke.log
for wl in taylorgreen64 drycblles256 drycblles512 moser600 gabls1_1024; do
  echo "== bench $wl"; timeout -k 10 600 python bench.py --workload $wl --steps 20 --warmup 3 > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err || { tail -5 $OUT/bench_$wl.err; exit 3; }
  cat $OUT/bench_$wl.json
done
echo "== bench drycblles512 unfused"; timeout -k 10 600 python bench.py --workload drycblles512 --steps 10 --warmup 3 --unfused --no-cpu-baseline > $OUT/bench_drycblles512_unfused.json 2> $OUT/bench_unfused.err || exit 4
cat $OUT/bench_drycblles512_unfused.json
for wl in slab8of512 gabls1_slab8; do
  echo "== bench $wl (one rank's share, run alone)"; timeout -k 10 600 python bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err || { tail -5 $OUT/bench_$wl.err; exit 3; }
  cat $OUT/bench_$wl.json
done
echo "== rocprofv3 kernel stats (moser600)"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_moser -- python3 bench.py --workload moser600 --steps 5 --warmup 2 --no-cpu-baseline --no-power-sample > $OUT/prof_moser_bench.json 2> $OUT/prof_moser.err || { tail -5 $OUT/prof_moser.err; exit 5; }
echo "== rocprofv3 kernel stats (drycblles512)"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 bench.py --workload drycblles512 --steps 5 --warmup 2 --no-cpu-baseline --no-power-sample > $OUT/prof_bench.json 2> $OUT/prof.err || { tail -5 $OUT/prof.err; exit 5; }
find $OUT/prof -name "*kernel_stats*.csv" | head -3
f=$(find $OUT/prof -name "*kernel_stats*.csv" | head -1); [ -n "$f" ] && head -25 "$f"
