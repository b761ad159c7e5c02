#!/bin/bash
# one engine option over a list of values on a workload's pre-screened leg       usage: opt_sweep.sh <tag> <workload> <option> <values...>
set -o pipefail
tag=$1; wl=$2; opt=$3; shift 3
out=gpurun_out/${tag}_${wl}_${opt}_sweep.log; : > $out
for rep in 1 2; do
for v in "$@"; do
  timeout -k 10 200 python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --opt $opt=$v > gpurun_out/${tag}_os.json 2> gpurun_out/${tag}_os.err || { echo "FAILED: $opt=$v" >> $out; tail -3 gpurun_out/${tag}_os.err >> $out; continue; }
  python - "$opt=$v" >> $out <<PY
import json, sys
b=json.loads(open('gpurun_out/${tag}_os.json').read().strip().splitlines()[-1])
p=b.get('exact_prescreen') or {}
print(f"rep $rep {sys.argv[1]:28s} direct {b['value']:9.0f} q/s | pre-screened {p.get('value',0):9.0f} q/s  {p.get('ms_per_step',0):7.2f} ms  identical {p.get('identical_to_direct')}  {p.get('breakdown_ms_per_step')}")
PY
done
done
cat $out
