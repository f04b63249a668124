#!/bin/bash
# scripts/gpu_r3a.sh -- round 3, call a: the row-alignment diagnosis of the fused 2i5+smag2 kernel.
# drycblles 512^3 with igc = 3 (reference default) and igc = 16 (rows = 34 whole 128-byte lines, istart on a line), for the
# shipped kernel, the memory-only and the arithmetic-only diagnostic builds; FETCH/WRITE of the shipped kernel at both.
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r3a
mkdir -p $OUT
echo "== pytest (new igc=16 parity test + marching parity on the GPU)"
timeout -k 10 600 python -m pytest tests/test_parity.py -m gpu -q -x -k "sixteen_ghost or fused_rhs or marching" > $OUT/pytest.log 2>&1; rc=$?
tail -3 $OUT/pytest.log
[ $rc -ne 0 ] && { echo "pytest failed: stopping"; exit $rc; }
run() {  # name, lib ("" = default), extra args...
  local name=$1 lib=$2; shift 2
  MHH_LIB=${lib:+$PWD/$lib} timeout -k 10 300 python bench.py --workload drycblles512 --steps 20 --warmup 3 --no-cpu-baseline --no-fma-line "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err || { echo "bench $name failed"; tail -3 $OUT/bench_$name.err; return 1; }
  python - "$OUT/bench_$name.json" "$name" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print("%-28s ms/step %7.3f  rhs ms %7.3f  rhs-frac %.3f  pres ms %s" % (sys.argv[2], d["ms_per_step"], d["roofline"]["ms_per_launch"], d["roofline"]["frac"], d.get("pressure",{}).get("ms")))
PY
}
run default_igc3 "" && run default_igc16 "" --igc 16 && MHH_MARCH_HX=3 run default_igc16_hx3 "" --igc 16 \
 && run nomath_igc3 microhh_amd/variants/libmhh_hip_nomath.so && run nomath_igc16 microhh_amd/variants/libmhh_hip_nomath.so --igc 16 \
 && run nomem_igc3 microhh_amd/variants/libmhh_hip_nomem.so && run nomem_igc16 microhh_amd/variants/libmhh_hip_nomem.so --igc 16 || exit 3
echo "== traffic, igc = 3"
ONLY_DEFAULT=1 bash scripts/gpu_traffic.sh r3a_traffic_igc3 drycblles512 rhs25_march
echo "== traffic, igc = 16"
ONLY_DEFAULT=1 BENCH_ARGS="--igc 16" bash scripts/gpu_traffic.sh r3a_traffic_igc16 drycblles512 rhs25_march
