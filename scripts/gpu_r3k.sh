#!/bin/bash
# slab path with the x stages in LDS: GPU tests + per-rank stage times (rank 0 of 8 on 512^3, exchanges skipped)
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r3k; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_parity.py tests/test_slab_gpu_ranks.py tests/test_cpp_host.py -m gpu -q -x -k "slab or pres or host" > $OUT/pytest.log 2>&1; rc=$?
tail -3 $OUT/pytest.log
[ $rc -ne 0 ] && { echo "pytest failed: stopping"; exit $rc; }
for cfg in "1 1" "0 1" "1 4" "0 4"; do set -- $cfg
  echo "== MHH_PRES_SLAB_LDS=$1 MHH_PRES_CHUNKS=$2"
  MHH_PRES_SLAB_LDS=$1 MHH_PRES_CHUNKS=$2 timeout -k 10 300 python scripts/slab_stage_timing.py 8 512 2>&1 | grep -v amdgpu.ids | tee $OUT/slab_stage_lds$1_chunks$2.txt
done
