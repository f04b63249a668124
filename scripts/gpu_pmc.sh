#!/bin/bash
# scripts/gpu_pmc.sh -- PMC counter passes (separate runs, --kernel-trace only) for the kernels of one bench workload.
set -o pipefail
TAG=${1:-pmc}; WL=${2:-drycblles512}
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
i=0
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" "TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU_MFMA_F64 SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $OUT/pmc$i -- python3 bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline --no-fma-line > $OUT/pmc$i.json 2> $OUT/pmc$i.err || { echo "pmc pass $i ($pmc) failed"; tail -2 $OUT/pmc$i.err; }
done
python - $OUT <<'PY'
import csv, glob, sys, collections
out=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
def short(k):
    for key,name in (("pres_in_fftx","pres_lds_in"),("pres_ysolve","pres_lds_ysolve"),("pres_ifftx_out","pres_lds_out"),("visc_march","viscmarch"),("rhs44_march","march4"),("rhs25_march","march"),("Rhs25","Rhs25cell"),("Viscosity","Visc"),("tdma","tdma"),("PresIn","PresIn"),("PresOut","PresOut"),("unpack","unpack"),("Rhs44","Rhs44"),("Rhs22","Rhs22"),("hdma","hdma")):
        if key in k: return name
    return None
for f in glob.glob(out+"/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        s = short(r["Kernel_Name"])
        if s: agg[s][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    print(k, {c: round(sum(x)/len(x)) for c,x in sorted(v.items())})
PY
