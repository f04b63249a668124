#!/bin/bash
# scripts/gpu_benchlines.sh <tag> -- the bench lines of every workload + the default invocation (plain and under rocprofv3 --kernel-trace --stats)
# + kernel statistics of moser600: the part of gpu_final.sh that depends on bench.py only (no tests, no PMC passes, no fuzz).
set -o pipefail
export TMPDIR=/tmp
TAG=${1:-r3z}; OUT=gpurun_out/$TAG; mkdir -p $OUT
for wl in taylorgreen64 drycblles256 drycblles512 moser600 gabls1_1024; do
  timeout -k 10 600 python bench.py --workload $wl --steps 20 --warmup 3 > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err || { tail -5 $OUT/bench_$wl.err; exit 3; }
  echo "$wl done"
done
timeout -k 10 600 python bench.py --workload drycblles512 --steps 10 --warmup 3 --unfused --no-cpu-baseline > $OUT/bench_drycblles512_unfused.json 2> $OUT/bench_unfused.err || exit 4
for wl in slab8of512 gabls1_slab8; do
  timeout -k 10 600 python bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err || { tail -5 $OUT/bench_$wl.err; exit 3; }
done
timeout -k 10 300 python bench.py --igc 16 --no-cpu-baseline > $OUT/bench_drycblles512_igc16.json 2> $OUT/bench_igc16.err || tail -3 $OUT/bench_igc16.err
echo "lines done"
timeout -k 10 600 python bench.py > $OUT/bench_default_plain.json 2> $OUT/bench_default_plain.err || tail -3 $OUT/bench_default_plain.err
rm -rf $OUT/prof_default $OUT/prof_moser
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_default -- python3 bench.py --no-power-sample > $OUT/bench_default_invocation.json 2> $OUT/bench_default_invocation.err || tail -3 $OUT/bench_default_invocation.err
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_moser -- python3 bench.py --workload moser600 --steps 5 --warmup 2 --no-cpu-baseline --no-power-sample > $OUT/prof_moser_bench.json 2> $OUT/prof_moser.err || tail -5 $OUT/prof_moser.err
python - $OUT <<'PY'
import json, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
    try: d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    print("%-36s step %7.3f ms  rhs %6.3f (frac %.3f, traffic %s)  pres %s  self-check ratio %s" % (f.split("/")[-1], d["ms_per_step"], d["roofline"]["ms_per_launch"], d["roofline"]["frac"],
          d["roofline"].get("traffic"), (d.get("pressure") or {}).get("ms"), (d.get("self_check") or {}).get("ratio")))
PY
