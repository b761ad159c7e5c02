#!/bin/bash
# fabric bytes (FETCH_SIZE x 2, gfx950 correction) and time of the fp16 tile kernel under several builds of the library, one rocprofv3 --pmc pass each:
# usage (GPU box, repo root): tools/fetch_libs.sh "<ab/lib1.so ab/lib2.so ...>" "<f16_forms.py args>"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in $1; do
  n=$(basename $lib .so); O=gpurun_out/fetch_$n; rm -rf $O
  export NWKNN_LIB_PATH=$PWD/$lib
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O -- python3 tools/f16_forms.py $2 > $O.log 2>&1 || { tail -3 $O.log; exit 1; }
  python3 - <<PY
import csv, glob
f = glob.glob("$O/*/*_counter_collection.csv")[0]
tot = 0.0; ms = 0.0; n = 0
for r in csv.DictReader(open(f)):
    if "nwk_tile_kernel_f16" in r["Kernel_Name"]:
        tot += float(r["Counter_Value"]); ms += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6; n += 1
print(f"$n: {n} launches, fabric {tot*1024*2/1e9/n:7.1f} GB per launch, {ms/n:7.2f} ms per launch ({tot*1024*2/1e9/(ms/1e3)/1e3:.2f} TB/s)")
PY
  rm -rf $O
done
