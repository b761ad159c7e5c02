#!/bin/bash
# usage: tools/mk_variant_lib.sh <out-name> <file.hip> "<extra hipcc flags, e.g. -DX_PRIO=0>"
# Builds ab/<out-name>.so = the current objects with ONE kernel file recompiled with extra flags, for interleaved A/B runs of two builds in
# one GPU session (NWKNN_LIB_PATH; tools/ab_forms.sh).  The package build must be current (python -m neighborhoodwatch_amd.build).
set -e
out=$1; f=$2; flags=$3
root=$(cd "$(dirname "$0")/.." && pwd)
c=$root/neighborhoodwatch_amd/csrc
mkdir -p $root/ab /tmp/ab_$out
hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -fhip-fp32-correctly-rounded-divide-sqrt -ffp-contract=off $flags -I$c -x hip -c $c/$f -o /tmp/ab_$out/obj.o
objs=""
for o in $c/build/*.o; do
  if [ "$(basename $o .o)" != "$(basename $f .hip)" ]; then objs="$objs $o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/ab_$out/obj.o -o $root/ab/$out.so
echo $root/ab/$out.so
