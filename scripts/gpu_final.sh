#!/bin/bash
# scripts/gpu_final.sh <tag> -- the evidence set of a round, on the final sources: gpu_check.sh (tests, smoke, bench lines, kernel stats), the
# default bench invocation under rocprofv3 --kernel-trace --stats, stamped traffic / VALU records of every GPU workload, the igc = 16 line,
# the clock / power probe, the fuzz runs. Everything lands under gpurun_out/<tag>*; copy what is to be judged into profiles/.
set -o pipefail
export TMPDIR=/tmp
TAG=${1:-r3z}
OUT=gpurun_out/$TAG; mkdir -p $OUT
bash scripts/gpu_check.sh $TAG > $OUT/check.log 2>&1; echo "gpu_check rc=$?"; grep -E "passed|failed|smoke ok" $OUT/check.log | head -5
echo "== default invocation under rocprofv3"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_default -- python3 bench.py --no-power-sample > $OUT/bench_default_invocation.json 2> $OUT/bench_default_invocation.err || tail -3 $OUT/bench_default_invocation.err
timeout -k 10 600 python bench.py > $OUT/bench_default_plain.json 2> $OUT/bench_default_plain.err || tail -3 $OUT/bench_default_plain.err
echo "== igc = 16"
timeout -k 10 300 python bench.py --igc 16 --no-cpu-baseline > $OUT/bench_drycblles512_igc16.json 2> $OUT/bench_igc16.err || tail -3 $OUT/bench_igc16.err
echo "== traffic + VALU records"
ONLY_DEFAULT=1 bash scripts/gpu_traffic.sh ${TAG}_t512 drycblles512 rhs25_march
ONLY_DEFAULT=1 BENCH_ARGS="--igc 16" bash scripts/gpu_traffic.sh ${TAG}_t512igc16 drycblles512 rhs25_march
ONLY_DEFAULT=1 bash scripts/gpu_traffic.sh ${TAG}_t256 drycblles256 rhs25_march
ONLY_DEFAULT=1 bash scripts/gpu_traffic.sh ${TAG}_tmoser moser600 rhs44_march
ONLY_DEFAULT=1 bash scripts/gpu_traffic.sh ${TAG}_tgabls gabls1_1024 rhs25_march
echo "== clock / power"
bash scripts/gpu_clock_probe.sh > $OUT/clock_power.txt 2>&1; tail -40 $OUT/clock_power.txt | grep -i "sclk\|power\|launches" | head -20
echo "== fuzz"
timeout -k 10 600 python scripts/experiments/parity_fuzz.py 400 31 > $OUT/parity_fuzz.log 2>&1; tail -1 $OUT/parity_fuzz.log
timeout -k 10 600 python scripts/experiments/pres_fuzz.py 150 13 lds > $OUT/pres_fuzz_lds.log 2>&1; tail -1 $OUT/pres_fuzz_lds.log
timeout -k 10 600 python scripts/experiments/pres_fuzz.py 120 14 > $OUT/pres_fuzz_staged.log 2>&1; tail -1 $OUT/pres_fuzz_staged.log
MHH_PRES_CHUNKS=1 timeout -k 10 300 python scripts/slab_stage_timing.py 8 512 > $OUT/slab_stage_timing.txt 2>&1; grep "full step" $OUT/slab_stage_timing.txt
