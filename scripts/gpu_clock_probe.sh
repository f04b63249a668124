#!/bin/bash
# scripts/gpu_clock_probe.sh [lib] -- engine clock, memory clock and socket power sampled by rocm-smi while bench.py loops
# (drycblles512, 400 steps in the background), for the power-limit question: does the chip hold its clock under this kernel?
export TMPDIR=/tmp
OUT=gpurun_out/clock; mkdir -p $OUT
LIB=$1
rocm-smi --showclocks --showpower --showmaxpower > $OUT/idle.txt 2>&1
for mode in rhs visc pres; do
  MHH_LIB=${LIB:+$PWD/$LIB} python scripts/experiments/rhs_loop.py $mode 7 > $OUT/loop_$mode.txt 2> $OUT/loop_$mode.err &
  BP=$!
  sleep 4.5
  for n in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>&1 | grep -i "sclk\|mclk\|power\|fclk" >> $OUT/load_$mode.txt; echo "--" >> $OUT/load_$mode.txt; sleep 0.4; done
  wait $BP
done
echo "== idle"; grep -i "sclk\|mclk\|power" $OUT/idle.txt | head -8
for mode in rhs visc pres; do echo "== under load ($mode)"; cat $OUT/loop_$mode.txt; head -16 $OUT/load_$mode.txt; done
