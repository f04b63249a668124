#!/bin/bash
# scripts/gpu_visc_ab.sh [tag] [rounds] -- exec_viscosity A/B on ONE box: the default library and every build under
# microhh_amd/variants/, round-robin; ms per launch of Diff::exec_viscosity (marching kernel + the evisc halo) on drycblles 512^3.
set -o pipefail
export TMPDIR=/tmp
TAG=${1:-viscab}; R=${2:-3}
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_parity.py tests/test_golden.py -m gpu -q -x -k "visc or smag or sixteen or sqrt" > $OUT/pytest.log 2>&1; rc=$?
tail -1 $OUT/pytest.log
[ $rc -ne 0 ] && { echo "pytest failed: stopping"; tail -30 $OUT/pytest.log; exit $rc; }
NAMES="default $(ls microhh_amd/variants/*.so 2>/dev/null | xargs -n1 basename 2>/dev/null | sed 's/libmhh_hip_//; s/\.so//')"
for r in $(seq 1 $R); do
  for name in $NAMES; do
    if [ "$name" = default ]; then unset MHH_LIB; else export MHH_LIB=$PWD/microhh_amd/variants/libmhh_hip_$name.so; fi
    timeout -k 10 200 python scripts/experiments/rhs_loop.py visc 4 > $OUT/visc_${name}_$r.txt 2> $OUT/visc_${name}_$r.err || { echo "$name failed"; tail -3 $OUT/visc_${name}_$r.err; exit 3; }
    echo "$name $r: $(cat $OUT/visc_${name}_$r.txt)"
  done
done
