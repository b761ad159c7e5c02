#!/bin/bash
# usage: ab_libs3.sh workload steps opts rounds lib...   -- interleaved comparison of several builds in one session
w=$1; s=$2; o=$3; R=$4; shift 4
for r in $(seq 1 $R); do
  for lib in "$@"; do
    NWKNN_LIB_PATH=$PWD/$lib timeout -k 10 400 python bench.py --workload $w --steps $s --warmup 1 --no-cpu-baseline --opt $o 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$w $o $lib', round(d['value']), 'q/s; kernel', round(d['roofline']['achieved'],1), 'TF tile_ms', round(d['breakdown_ms_per_step']['tile'],1))"
  done
done
