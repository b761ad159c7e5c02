#!/bin/bash
# ab/f32_<name>.so = the current library with nwk_tile_f32.o replaced by a variant build (same-box A/B of the fp32 tile kernel):
#   head    : the file as committed at <rev> (default HEAD~: per-instruction VGPR offsets), stamp branches kept  (round 2/3's kernel: 168 VGPRs, 7 spills)
#   new     : the working tree's file (one per-lane offset + scalar row offsets), stamp code compiled out        (5 spills)
#   new_dbg : the working tree's file with the stamp branches kept (-DNWK_KEEP_DBG)                              (8 spills)
#   old_nodbg: <rev>'s file with the stamp code compiled out                                                     (13 spills)
set -e
rev=${1:-HEAD}
root=$(cd "$(dirname "$0")/.." && pwd); c=$root/neighborhoodwatch_amd/csrc
mkdir -p $root/ab /tmp/f32v
FL="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -fhip-fp32-correctly-rounded-divide-sqrt -ffp-contract=off -I$c -x hip"
git -C $root show $rev:neighborhoodwatch_amd/csrc/nwk_tile_f32.hip > /tmp/f32v/old.hip
hipcc $FL -DNWK_KEEP_DBG -c /tmp/f32v/old.hip -o /tmp/f32v/head.o &
hipcc $FL -c /tmp/f32v/old.hip -o /tmp/f32v/old_nodbg.o &
hipcc $FL -c $c/nwk_tile_f32.hip -o /tmp/f32v/new.o &
hipcc $FL -DNWK_KEEP_DBG -c $c/nwk_tile_f32.hip -o /tmp/f32v/new_dbg.o &
wait
objs=""
for o in $c/build/*.o; do [ "$(basename $o)" != "nwk_tile_f32.o" ] && objs="$objs $o"; done
for v in head old_nodbg new new_dbg; do
  hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/f32v/$v.o -o $root/ab/f32_$v.so
  echo $root/ab/f32_$v.so
done
