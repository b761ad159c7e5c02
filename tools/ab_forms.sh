#!/bin/bash
# usage: ab_forms.sh "<lib1 lib2 ...>" "<f16_forms.py args>" [rounds]  -- tools/f16_forms.py under each library in turn, interleaved
libs=$1; args=$2; R=${3:-2}
for r in $(seq 1 $R); do
  for lib in $libs; do
    echo "== $lib (round $r)"
    NWKNN_LIB_PATH=$PWD/$lib timeout -k 10 300 python tools/f16_forms.py $args 2>&1 | grep -v amdgpu.ids
  done
done
