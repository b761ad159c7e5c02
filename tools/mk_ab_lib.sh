#!/bin/bash
# usage: tools/mk_ab_lib.sh <git-rev> <file.hip> <out-name>
# Builds ab/<out-name>.so = the current objects with ONE kernel file taken from <git-rev> (compiled against the current headers),
# for interleaved A/B runs of two builds in one GPU session (NWKNN_LIB_PATH, tools/ab_libs.sh).
set -e
rev=$1; f=$2; out=$3
root=$(cd "$(dirname "$0")/.." && pwd)
c=$root/neighborhoodwatch_amd/csrc
mkdir -p $root/ab /tmp/ab_$out
git -C $root show $rev:neighborhoodwatch_amd/csrc/$f > /tmp/ab_$out/$f
hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -fhip-fp32-correctly-rounded-divide-sqrt -ffp-contract=off -I$c -x hip -c /tmp/ab_$out/$f -o /tmp/ab_$out/obj.o
objs=""
for o in $c/build/*.o; do
  if [ "$(basename $o .o)" != "$(basename $f .hip)" ]; then objs="$objs $o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/ab_$out/obj.o -o $root/ab/$out.so
echo $root/ab/$out.so
