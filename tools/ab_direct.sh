#!/bin/bash
# usage: ab_direct.sh "<B1> <B2>" libA libB [rounds] -- direct fp32 search (100k x B, d=1024), two libraries interleaved
R=${4:-2}
for B in $1; do
  for r in $(seq 1 $R); do
    for lib in $2 $3; do
      NWKNN_LIB_PATH=$PWD/$lib timeout -k 10 400 python tools/chunk_times.py $B 0 2>/dev/null | tail -1 | python -c "
import sys,ast
d=ast.literal_eval(sys.stdin.read())
print('B=$B $lib', 'tile', d['tile_ms'], 'total', d['total_ms'], 'tile TF', round(d['tile_flops']/d['tile_ms']/1e9,2))"
    done
  done
done
