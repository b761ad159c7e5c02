#!/bin/bash
# same-box A/B of four builds of the fp32 tile kernel (tools/mk_f32_variants.sh), C3 shape, direct search, interleaved, 2 rounds
mkdir -p gpurun_out
out=gpurun_out/r04c_f32_variants.log
: > $out
for r in 1 2; do
  for v in head old_nodbg new new_dbg; do
    NWKNN_LIB_PATH=$PWD/ab/f32_$v.so timeout -k 10 300 python tools/chunk_times.py 10000000 0 2>/dev/null | tail -1 | python -c "
import sys,ast
d=ast.literal_eval(sys.stdin.read())
print('round $r $v', 'tile', d['tile_ms'], 'total', d['total_ms'], 'tile TF', round(d['tile_flops']/d['tile_ms']/1e9,2))" >> $out
    tail -1 $out
  done
done
