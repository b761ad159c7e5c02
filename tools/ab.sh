#!/bin/bash
# usage: ab.sh workload steps warmup opt1 opt2 ...   (each opt like persistent=0)
w=$1; s=$2; wu=$3; shift 3
for o in "$@"; do
  timeout -k 10 400 python bench.py --workload $w --steps $s --warmup $wu --no-cpu-baseline --opt $o 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$w $o', round(d['value']), 'q/s', round(d['pairwise_tflops'],1), 'TF e2e; kernel', round(d['roofline']['achieved'],1), 'TF', {k:(round(v,2) if isinstance(v,float) else v) for k,v in d['breakdown_ms_per_step'].items()})"
done
