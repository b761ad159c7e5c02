#!/bin/bash
# the chunk walk's knobs (list capacity, dense first chunk, growth) on the pre-screened legs of a workload      usage: walk_sweep.sh <tag> <workload>
set -o pipefail
tag=${1:-ws}; wl=${2:-c3shard}
out=gpurun_out/${tag}_${wl}_walk_sweep.log; : > $out
for cfg in "" "capacity=2048" "capacity=4096" "capacity=2048 first_chunk=1024" "capacity=4096 first_chunk=2048" "capacity=4096 first_chunk=2048 chunk_growth=7" "capacity=4096 first_chunk=1024 chunk_growth=7" "capacity=2048 chunk_growth=5" "capacity=8192 first_chunk=4096 chunk_growth=7"; do
  opts=""; for o in $cfg; do opts="$opts --opt $o"; done
  timeout -k 10 200 python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline $opts > gpurun_out/${tag}_ws.json 2> gpurun_out/${tag}_ws.err || { echo "FAILED: $cfg" >> $out; tail -3 gpurun_out/${tag}_ws.err >> $out; continue; }
  python - "$cfg" >> $out <<PY
import json, sys
b=json.loads(open('gpurun_out/${tag}_ws.json').read().strip().splitlines()[-1])
p=b.get('exact_prescreen') or {}
print(f"{sys.argv[1] or 'default':60s} direct {b['value']:9.0f} q/s | pre-screened {p.get('value',0):9.0f} q/s  {p.get('ms_per_step',0):7.2f} ms  {p.get('breakdown_ms_per_step')}")
PY
done
cat $out
