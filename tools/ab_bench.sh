#!/bin/bash
# bench.py lines of one workload under several builds of the library, interleaved: tools/ab_bench.sh "<ab/lib1.so ...>" "<bench.py args>" [rounds]
# prints value, pre-screened value and the pre-screen tile kernel's TF per run
libs=$1; args=$2; R=${3:-2}
for r in $(seq 1 $R); do
  for lib in $libs; do
    NWKNN_LIB_PATH=$PWD/$lib timeout -k 10 600 python bench.py $args --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.readlines()[-1]); p = d.get('exact_prescreen') or {}
print('$lib round $r: %.0f q/s direct (%.2f ms)' % (d['value'], d['ms_per_step']), '| pre-screened %.0f q/s (%.2f ms/step; tile %.1f TF %s; tile %.2f rescore %.2f compact %.2f prep %.2f ms)' % (
      p['value'], p['ms_per_step'], p['roofline']['achieved'], p['roofline']['kernel'], *[p['breakdown_ms_per_step'][k] for k in ('tile', 'rescore', 'compact', 'prep')]) if p else '')"
  done
done
