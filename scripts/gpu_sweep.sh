#!/bin/bash
# scripts/gpu_sweep.sh -- tuning sweep: variant builds of the same sources + PMC counters of the baseline.
set -o pipefail
TAG=${1:-sweep}
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
echo "== pytest -m gpu"; timeout -k 10 900 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; rc=$?
tail -4 $OUT/pytest_gpu.log
if [ $rc -gt 1 ]; then echo "pytest crashed (rc=$rc): stopping"; exit $rc; fi
for v in "" $(ls microhh_amd/variants/*.so 2>/dev/null); do
  name=$(basename "${v:-default}" .so)
  echo "== bench $name"
  MHH_LIB=${v:+$PWD/$v} timeout -k 10 300 python bench.py --workload drycblles512 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_$name.json 2> $OUT/bench_$name.err || { tail -3 $OUT/bench_$name.err; continue; }
  python - "$OUT/bench_$name.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print("   ms/step %.2f  rhs ms %.3f  value %.3e" % (d["ms_per_step"], d["roofline"]["ms_per_launch"], d["value"]))
PY
done
if [ -n "$SKIP_PMC" ]; then exit 0; fi
echo "== PMC passes (baseline)"
rocprofv3 -L > $OUT/counters_list.txt 2>&1 || true
i=0
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAVES"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $OUT/pmc$i -- python3 bench.py --workload drycblles512 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc$i.json 2> $OUT/pmc$i.err || { echo "pmc pass $i failed"; tail -3 $OUT/pmc$i.err; }
done
python - $OUT <<'PY'
import csv, glob, sys, collections
out=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out+"/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        short = "Rhs25" if "Rhs25" in k else "Visc" if "Viscosity" in k else "tdma" if "tdma" in k else "PresIn" if "PresIn" in k else "PresOut" if "PresOut" in k else None
        if short: agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    print(k, {c: sum(x)/len(x) for c,x in v.items()})
PY
