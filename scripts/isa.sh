#!/bin/bash
# scripts/isa.sh <object.o> <kernel-substring> [out-prefix] -- unbundle the gfx950 code object of a hipcc object file, disassemble it
# and print the loop statistics of the matching kernels (scripts/isa_loop_stats.py) + their register / scratch / LDS metadata.
set -e
OBJ=$(readlink -f $1); PAT=$2; PRE=${3:-/tmp/isa/$(basename $1 .o)}
mkdir -p $(dirname $PRE); T=$(mktemp -d); cp $OBJ $T/x.o
(cd $T && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading x.o > /dev/null)
mv $T/x.o.0.hipv4-amdgcn-amd-amdhsa--gfx950 $PRE.co; rm -rf $T
/opt/rocm/lib/llvm/bin/llvm-objdump -d $PRE.co > $PRE.s
python3 $(dirname $0)/isa_loop_stats.py $PRE.s "$PAT"
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $PRE.co | python3 -c "
import sys,re
txt=sys.stdin.read()
for blk in txt.split('- .agpr_count')[1:]:
    name=re.search(r'\.name:\s+(\S+)',blk)
    if name and '$PAT' in name.group(1):
        g=lambda k:(re.search(k+r':\s+(\S+)',blk) or [None,'?'])[1]
        print('   meta: vgpr',g('.vgpr_count'),'sgpr',g('.sgpr_count'),'sgpr_spill',g('.sgpr_spill_count'),'vgpr_spill',g('.vgpr_spill_count'),'scratch',g('.private_segment_fixed_size'),'lds',g('.group_segment_fixed_size'))
"
