#!/bin/bash
# scripts/gpu_pmc2.sh <tag> <workload> <kernel-pattern> -- PMC counter groups (one rocprofv3 pass per group, --kernel-trace only)
# for the default library and every build under microhh_amd/variants/ (ONLY="a b" limits them; "default" = the shipped one).
# Groups: the lines of $GROUPS_FILE (default scripts/pmc_groups.txt). Prints per build the per-launch average of every counter.
set -o pipefail
TAG=${1:-pmc2}; WL=${2:-drycblles512}; PAT=${3:-rhs25_march}
GROUPS_FILE=${GROUPS_FILE:-scripts/pmc_groups.txt}
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
rocprofv3 --list-avail > $OUT/avail.txt 2>&1 || true
LIBS="default $(ls microhh_amd/variants/*.so 2>/dev/null | xargs -n1 basename 2>/dev/null | sed 's/libmhh_hip_//; s/\.so//')"
[ -n "$ONLY" ] && LIBS="$ONLY"
for name in $LIBS; do
  if [ "$name" = default ]; then unset MHH_LIB; else export MHH_LIB=$PWD/microhh_amd/variants/libmhh_hip_$name.so; fi
  i=0
  while read -r grp; do
    [ -z "$grp" ] && continue; case "$grp" in \#*) continue;; esac
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/$name/g$i -- python3 bench.py --workload $WL --steps 4 --warmup 1 --no-cpu-baseline --no-fma-line --no-power-sample $BENCH_ARGS > $OUT/$name.g$i.json 2> $OUT/$name.g$i.err || { echo "$name group $i ($grp) failed"; tail -2 $OUT/$name.g$i.err; }
  done < $GROUPS_FILE
  python3 - $OUT/$name "$PAT" "$name" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("== %s (%s)" % (sys.argv[3], sys.argv[2]))
for c, x in sorted(agg.items()):
    print("   %-40s %.6g" % (c, sum(x)/len(x)))
PY
done
