#!/bin/bash
# usage: opts_ab.sh B "<opt-set-A>" "<opt-set-B>" [rounds]  -- pre-screened search (100k x B, d=1024) under two option sets, interleaved
B=$1; R=${4:-2}
for r in $(seq 1 $R); do
  for o in "$2" "$3"; do
    NWKNN_DEFAULT_OPTS="$o" timeout -k 10 300 python tools/chunk_times.py $B 1 2>/dev/null | tail -1 | python -c "
import sys,ast
d=ast.literal_eval(sys.stdin.read())
print('B=$B', '$o', 'tile', d['tile_ms'], 'rescore', d['rescore_ms'], 'total', d['total_ms'], 'tile TF', round(d['tile_flops']/d['tile_ms']/1e9,1))"
  done
done
