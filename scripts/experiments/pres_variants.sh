#!/bin/bash
# time the pressure forms of every build under microhh_amd/variants/ (and the default library) -- pres_lds_probe.py per library
set -o pipefail
SPEC=${1:-drycblles:512:512:512}
for v in "" $(ls microhh_amd/variants/*.so 2>/dev/null); do
  name=$(basename "${v:-default}" .so)
  for kc in ${KCS:-32}; do
    echo "== $name kc=$kc"
    MHH_PRES_LDS_KC=$kc MHH_LIB=${v:+$PWD/$v} timeout -k 10 300 python scripts/experiments/pres_lds_probe.py $SPEC 2>&1 | grep "rel diff" || echo failed
  done
done
