#!/bin/bash
# the C4 per-GPU share under rocprofv3 --kernel-trace --stats at HEAD (tile_launches = HIP launches): kernel stats vs the bench line's HIP events
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=${1:-r04t}; O=gpurun_out/prof_$T; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c4gpu -- python3 bench.py --workload c4gpu --steps 1 --warmup 0 --no-cpu-baseline > $O/${T}_c4gpu_bench_under_rocprof.json 2> $O/c4gpu_bench.err || exit 1
find $O/trace_c4gpu -name "*_kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${T}_c4gpu_kernel_stats.csv
find $O/trace_c4gpu -name "*_kernel_trace.csv" -delete
python3 - <<PY
import csv, json
O, T = "$O", "$T"
b = json.loads(open(f"{O}/{T}_c4gpu_bench_under_rocprof.json").read().strip().splitlines()[-1])
tot = 0.0; n = 0
for r in csv.DictReader(open(f"{O}/{T}_c4gpu_kernel_stats.csv")):
    if "nwk_tile_kernel" in r["Name"]:
        print(r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e6, "ms avg")
        tot += float(r["TotalDurationNs"]) / 1e6; n += int(r["Calls"])
rf = b["roofline"]
print("rocprofv3: %d tile launches, %.2f ms in total, %.2f ms avg over the %d filtered ones | bench line: launches %d, avg_launch_ms %.2f, tile %.2f ms, %s q/s, frac %.3f / %.3f" % (
      n, tot, 0.0, n - 1, rf["launches"], rf["avg_launch_ms"], b["breakdown_ms_per_step"]["tile"], round(b["value"]), rf["frac"], rf.get("frac_of_bare_loop", 0)))
PY
