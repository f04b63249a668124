#!/bin/bash
# scripts/gpu_pres4_prof.sh [tag] -- stage times of the pres_4 LDS form at moser600's grid + rocprofv3 kernel statistics of a short bench run
set -o pipefail
export TMPDIR=/tmp
TAG=${1:-pres4prof}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 300 python scripts/experiments/pres_lds_probe.py moser600:512:256:256 moser600:256:256:256 moser600:1024:512:128 2>&1 | tee $OUT/probe.txt || exit 3
cd /tmp && MHH_PRES_LDS=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o moser -- python3 $GRAFT_REPO_ROOT/bench.py --workload moser600 --steps 20 --warmup 3 --no-cpu-baseline --no-fma-line --no-power-sample > $GRAFT_REPO_ROOT/$OUT/bench.json 2> $GRAFT_REPO_ROOT/$OUT/bench.err || { tail -5 $GRAFT_REPO_ROOT/$OUT/bench.err; exit 4; }
cd $GRAFT_REPO_ROOT
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-150
