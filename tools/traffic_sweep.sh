#!/bin/bash
# FETCH_SIZE of the tile kernel for several super-tile shapes (one rocprofv3 --pmc pass each), workload $1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
w=${1:-c2}
for sh in 3 0 1 2 4 5 6; do
  rm -rf gpurun_out/sweep_$sh
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/sweep_$sh -- python3 bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline --opt super_q_shift=$sh > gpurun_out/sweep_$sh.json 2> gpurun_out/sweep_$sh.err
  python3 - <<PY
import csv, glob, json
f = glob.glob("gpurun_out/sweep_$sh/*/*_counter_collection.csv")[0]
tot = 0.0; ms = 0.0
for r in csv.DictReader(open(f)):
    if "tile_kernel" in r["Kernel_Name"] and "false" in r["Kernel_Name"]:
        tot += float(r["Counter_Value"]); ms += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
d = json.load(open("gpurun_out/sweep_$sh.json"))
print(f"super_q_shift=$sh ({1<<$sh} q-tiles x {64>>$sh} b-tiles): fetch(x2) {tot*1024*2/1e9:8.1f} GB  kernel {ms:8.2f} ms  e2e {d['pairwise_tflops']:.1f} TF")
PY
done
