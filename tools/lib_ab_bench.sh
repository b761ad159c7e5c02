#!/bin/bash
# same-box A/B of two library builds on bench workloads, alternating A B A B      usage: lib_ab_bench.sh <other lib> <tag> <workload> [steps]
set -o pipefail
other=$1; tag=${2:-ab}; wl=${3:-c5}; steps=${4:-2}
out=gpurun_out/${tag}_${wl}_lib_ab.log; : > $out
for rep in 1 2; do
  for which in A B; do
    lib=""; [ $which = B ] && lib=$other
    NWKNN_LIB_PATH=$lib timeout -k 10 300 python bench.py --workload $wl --steps $steps --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_${wl}_$which$rep.json 2> gpurun_out/${tag}_${wl}_$which$rep.err || { tail -5 gpurun_out/${tag}_${wl}_$which$rep.err; exit 1; }
    python - >> $out <<PY
import json
b=json.loads(open('gpurun_out/${tag}_${wl}_$which$rep.json').read().strip().splitlines()[-1])
rf=b.get('roofline',{}); p=b.get('exact_prescreen') or {}
print('$which$rep', '$wl', 'value', round(b['value'],1), 'frac', rf.get('frac'), 'pre', p.get('value'), (p.get('roofline') or {}).get('frac'))
PY
  done
done
cat $out
