#!/bin/bash
# same-box A/B of two library builds on the fp16 tile kernels: forms 4 and 5 at several row lengths, alternating A B A B
# usage: lib_ab.sh <other lib> <tag>      (A = the in-tree library, B = <other lib>)
set -o pipefail
other=$1; tag=${2:-ab}
out=gpurun_out/${tag}_lib_ab.log; : > $out
for d in 1024 1536 512 128; do
  mode=prescreen; [ $d = 1536 ] && mode=f16
  for rep in 1 2; do
    echo "== d=$d $mode  A (in-tree)" >> $out
    timeout -k 10 200 python tools/f16_forms.py "4 5" $d 4000000 32768 2 $mode 2>&1 | grep -i "form\|TF" >> $out || exit 1
    echo "== d=$d $mode  B ($other)" >> $out
    NWKNN_LIB_PATH=$other timeout -k 10 200 python tools/f16_forms.py "4 5" $d 4000000 32768 2 $mode 2>&1 | grep -i "form\|TF" >> $out || exit 1
  done
done
cat $out
