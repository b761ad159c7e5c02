#!/bin/bash
# usage: ab_libs_opt.sh workload steps warmup opts libA libB [rounds]   -- interleaved A/B of two builds with engine options
w=$1; s=$2; wu=$3; o=$4; A=$5; B=$6; R=${7:-2}
for r in $(seq 1 $R); do
  for lib in $A $B; do
    NWKNN_LIB_PATH=$PWD/$lib timeout -k 10 400 python bench.py --workload $w --steps $s --warmup $wu --no-cpu-baseline --opt $o 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$w $o $lib', round(d['value']), 'q/s; kernel', round(d['roofline']['achieved'],1), 'TF', {k:(round(v,1) if isinstance(v,float) else v) for k,v in d['breakdown_ms_per_step'].items() if k in ('tile','rescore','prep')})"
  done
done
