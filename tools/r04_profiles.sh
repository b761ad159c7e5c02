#!/bin/bash
# round 4's profiles: the default bench command under rocprofv3 --kernel-trace --stats + PMC passes (tools/profile_round.sh), then the
# C4 per-GPU share (1M queries) under --kernel-trace --stats and its three PMC passes (tools/pmc_workload.sh)
T=${1:-r04k}
bash tools/profile_round.sh $T || exit 1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_$T
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c4gpu -- python3 bench.py --workload c4gpu --steps 1 --warmup 0 --no-cpu-baseline > $O/${T}_c4gpu_bench_under_rocprof.json 2> $O/c4gpu_bench.err || exit 1
find $O/trace_c4gpu -name "*_kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${T}_c4gpu_kernel_stats.csv
find $O/trace_c4gpu -name "*_kernel_trace.csv" -delete
echo "c4gpu trace done"
bash tools/pmc_workload.sh c4gpu $T || exit 1
cp gpurun_out/pmcw_$T/${T}_c4gpu_pmc_summary.json $O/
find gpurun_out/pmcw_$T -name "*_counter_collection.csv" -delete; find gpurun_out/pmcw_$T -name "*_kernel_trace.csv" -delete
python3 - <<PY
import csv, json
O, T = "$O", "$T"
for w, kern in (("c3", "nwk_tile_kernel<0, false, 256, 16, 0, 3>"), ("c4gpu", "nwk_tile_kernel_f16x<0, 1>")):
    b = json.loads(open(f"{O}/{T}_{w}_bench_under_rocprof.json").read().strip().splitlines()[-1])
    rows = [r for r in csv.DictReader(open(f"{O}/{T}_{w}_kernel_stats.csv")) if kern.replace(" ", "") in r["Name"].replace(" ", "")]
    for r in rows:
        print(w, r["Name"][:60], "calls", r["Calls"], "avg ms", float(r["AverageNs"]) / 1e6, "| bench line avg_launch_ms", b["roofline"]["avg_launch_ms"], "kernel", b["roofline"]["kernel"], "frac", b["roofline"]["frac"], b["roofline"].get("frac_of_bare_loop"))
    if w == "c3":
        p = b["exact_prescreen"]; print("   pre", p["value"], p["roofline"]["avg_launch_ms"], p["roofline"]["frac"], p["roofline"].get("frac_of_bare_loop"), p["invariant"])
PY
