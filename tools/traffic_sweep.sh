#!/bin/bash
# FETCH_SIZE of the tile kernel for several super-tile shapes and walk orders (one rocprofv3 --pmc pass each): traffic_sweep.sh WORKLOAD [extra --opt]
# (pre-screened path: NOPRE= tools/traffic_sweep.sh c3 prescreen=1)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
w=${1:-c2}; extra=${2:-profile=1}
for cfg in ${CFGS:-3:0 3:4 4:0 4:4 2:4 1:4 5:0 2:0}; do
  sh=${cfg%%:*}; fl=${cfg##*:}
  rm -rf gpurun_out/sweep_${sh}_$fl
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/sweep_${sh}_$fl -- python3 bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline --no-secondary ${NOPRE---no-prescreen} --opt super_q_shift=$sh,super_walk_q=$((fl>0)),$extra > gpurun_out/sweep_${sh}_$fl.json 2> gpurun_out/sweep_${sh}_$fl.err
  python3 - <<PY
import csv, glob, json
f = glob.glob("gpurun_out/sweep_${sh}_$fl/*/*_counter_collection.csv")[0]
tot = 0.0; ms = 0.0
for r in csv.DictReader(open(f)):
    if "tile_kernel" in r["Kernel_Name"] and "true" not in r["Kernel_Name"].split("(")[0]:
        tot += float(r["Counter_Value"]); ms += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
d = json.load(open("gpurun_out/sweep_${sh}_$fl.json"))
print(f"$w $extra: {1<<$sh:2d} q-tiles x {64>>$sh:2d} b-tiles, {'q' if $fl else 'b'} fastest: fetch(x2) {tot*1024*2/1e9:8.1f} GB  kernel {ms:8.2f} ms  e2e {d['pairwise_tflops']:.1f} TF")
PY
  rm -rf gpurun_out/sweep_${sh}_$fl
done
