#!/bin/bash
# usage: ab_pre.sh workload steps opts...   -- prints direct and pre-screen numbers for each option set (one bench run each)
w=$1; s=$2; shift 2
for o in "$@"; do
  timeout -k 10 600 python bench.py --workload $w --steps $s --warmup 1 --no-cpu-baseline --opt $o 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
e=d.get('exact_prescreen')
print('$w $o direct', round(d['value']), 'q/s kernel', round(d['roofline']['achieved'],1), 'TF')
if e: print('$w $o PRE   ', round(e['value']), 'q/s kernel', round(e['roofline']['achieved'],1), 'TF identical', e['identical_to_direct'], {k:(round(v,1) if isinstance(v,float) else v) for k,v in e['breakdown_ms_per_step'].items()})"
done
