#!/bin/bash
# usage: ab_libs.sh workload steps warmup libA libB [rounds]   -- interleaved A/B of two builds in one session
w=$1; s=$2; wu=$3; A=$4; B=$5; R=${6:-2}
for r in $(seq 1 $R); do
  for lib in $A $B; do
    NWKNN_LIB_PATH=$PWD/$lib timeout -k 10 400 python bench.py --workload $w --steps $s --warmup $wu --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$w $lib', round(d['value']), 'q/s', round(d['pairwise_tflops'],1), 'TF e2e; kernel', round(d['roofline']['achieved'],1), 'TF')"
  done
done
