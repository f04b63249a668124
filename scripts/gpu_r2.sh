#!/bin/bash
# scripts/gpu_r2.sh -- round-2 GPU call: parity on the march kernels first, then timing of the default and variant builds.
set -o pipefail
TAG=${1:-r2a}
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
echo "== pytest parity/golden -m gpu"; timeout -k 10 900 python -m pytest tests/test_parity.py tests/test_golden.py -m gpu -q -x > $OUT/pytest_parity.log 2>&1; rc=$?
tail -6 $OUT/pytest_parity.log
if [ $rc -ne 0 ]; then echo "parity failed (rc=$rc): stopping"; exit $rc; fi
bash scripts/gpu_variants.sh $TAG drycblles512 || exit 3
echo "== fullsize"; timeout -k 10 900 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -x > $OUT/pytest_fullsize.log 2>&1; tail -4 $OUT/pytest_fullsize.log
