#!/bin/bash
# layout pass: non-temporal stores (and loads) against the default, interleaved on one box
mkdir -p gpurun_out; out=gpurun_out/r04s_prep_nt_ab.log; : > $out
for r in 1 2; do
  for lib in neighborhoodwatch_amd/libnwknn.so ab/prep_nt1.so ab/prep_nt2.so; do
    echo "## round $r $lib" >> $out
    NWKNN_LIB_PATH=$PWD/$lib python tools/prep_rate.py 4 2>/dev/null >> $out
  done
done
cat $out
