#!/bin/bash
# FETCH_SIZE of the tile kernel for one workload/options: tools/pmc_fetch.sh WORKLOAD [OPT]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
w=$1; o=${2:-profile=1}
rm -rf gpurun_out/pmc_$w
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_$w -- python3 bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline --opt $o > gpurun_out/pmc_$w.json 2> gpurun_out/pmc_$w.err
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/pmc_$w/*/*_counter_collection.csv")[0]
tot = 0.0; ms = 0.0; n = 0
for r in csv.DictReader(open(f)):
    if ("tile_kernel" in r["Kernel_Name"] or "maxsim" in r["Kernel_Name"]) and "false" in r["Kernel_Name"]:
        tot += float(r["Counter_Value"]); ms += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6; n += 1
print(f"$w $o: {n} launches, fetch(x2) {tot*1024*2/1e9:9.1f} GB in {ms:9.2f} ms = {tot*1024*2/1e9/ms:6.2f} TB/s")
PY
