#!/bin/bash
# per-launch durations of every kernel of one search's chunk walk, in launch order: tools/chunk_trace.sh TAG <chunk_times.py args>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
O=gpurun_out/chunktrace_$tag; rm -rf $O
rocprofv3 --kernel-trace --output-format csv -d $O -- python3 tools/chunk_times.py "$@" > $O.log 2>&1 || { tail -5 $O.log; exit 1; }
python3 - <<PY
import csv, glob
f = glob.glob("$O/*/*_kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the LAST search = everything after the last finalize-but-one
ends = [i for i, r in enumerate(rows) if "finalize" in r["Kernel_Name"]]
lo = ends[-2] + 1 if len(ends) > 1 else 0
t0 = int(rows[lo]["Start_Timestamp"])
for r in rows[lo:ends[-1] + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e6:9.3f} ms  +{(e - s) / 1e6:8.3f} ms  grid {r.get('Grid_Size', '?'):>10}  {r['Kernel_Name'].split('(')[0][:70]}")
print(f"search wall (first kernel start to finalize end): {(int(rows[ends[-1]]['End_Timestamp']) - t0) / 1e6:.3f} ms")
PY
rm -rf $O
