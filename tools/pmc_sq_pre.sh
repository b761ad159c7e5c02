#!/bin/bash
# SQ counters (MFMA busy, wave cycles, waits, clock) of the pre-screen tile kernel on one workload
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
w=${1:-c3}
rm -rf gpurun_out/pmcsq_$w
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES --output-format csv -d gpurun_out/pmcsq_$w -- python3 bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline --opt prescreen=1 > gpurun_out/pmcsq_$w.json 2> gpurun_out/pmcsq_$w.err || exit 1
python3 - <<PY
import csv, glob, json, collections
f = glob.glob("gpurun_out/pmcsq_$w/*/*_counter_collection.csv")[0]
acc = collections.defaultdict(float); ms = 0.0; n = 0; seen = set()
for r in csv.DictReader(open(f)):
    if "nwk_tile_kernel_f16" in r["Kernel_Name"]:
        acc[r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r["Dispatch_Id"] if "Dispatch_Id" in r else r["Start_Timestamp"])
        if key not in seen:
            seen.add(key); ms += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6; n += 1
clk = acc["GRBM_GUI_ACTIVE"] / 8 / (ms * 1e-3) / 1e9
out = {"workload": "$w", "kernel": "nwk_tile_kernel_f16s<0, false, 1>", "launches": n, "kernel_ms_total": ms, "counters": dict(acc),
       "effective_clock_ghz": clk,
       "mfma_busy_fraction_of_simd_cycles": acc["SQ_VALU_MFMA_BUSY_CYCLES"] / (acc["GRBM_GUI_ACTIVE"] / 8 * 256 * 4),
       "wave_residency": acc["SQ_WAVE_CYCLES"] * 4 / (acc["GRBM_GUI_ACTIVE"] / 8 * 256 * 4 * 2) if acc["SQ_WAVE_CYCLES"] else None,   # tallied in units of 4 cycles
       "source": "rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES (tools/pmc_sq_pre.sh)"}
json.dump(out, open("gpurun_out/pmcsq_${w}_summary.json", "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "counters"}, indent=1))
PY
