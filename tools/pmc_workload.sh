#!/bin/bash
# usage (GPU box, repo root): tools/pmc_workload.sh WORKLOAD TAG ["extra bench args"]
# Three rocprofv3 PMC passes (FETCH_SIZE | WRITE_SIZE | clock + MFMA busy; each its own run with --kernel-trace only) over
# `bench.py --workload WORKLOAD --steps 1 --warmup 0`, summarised per tile / MaxSim kernel into gpurun_out/pmcw_TAG/TAG_WORKLOAD_pmc_summary.json
# (fabric bytes = FETCH_SIZE x 2 per the gfx950 correction of MI355X_MICROARCH.md + WRITE_SIZE; clock = GRBM_GUI_ACTIVE / 8 / time).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
w=$1; T=$2; extra=$3; O=gpurun_out/pmcw_$T; mkdir -p $O
for grp in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES"; do
  n=$(echo $grp | cut -d' ' -f1)
  rm -rf $O/${w}_$n
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/${w}_$n -- python3 bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline $extra > $O/${w}_$n.json 2> $O/${w}_$n.err || exit 1
  echo "$w $n done"
done
python3 - <<PY
import collections, csv, glob, json
O, w, T = "$O", "$w", "$T"
def load(name):
    f = glob.glob(f"{O}/{w}_{name}/*/*_counter_collection.csv")[0]
    per = collections.defaultdict(lambda: {"c": collections.defaultdict(float), "ms": 0.0, "seen": set()})
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not ("tile_kernel" in k or "maxsim" in k or "rescore" in k):
            continue
        e = per[k]; e["c"][r["Counter_Name"]] += float(r["Counter_Value"])
        key = r.get("Dispatch_Id") or r["Start_Timestamp"]
        if key not in e["seen"]:
            e["seen"].add(key); e["ms"] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    return per
F, W, S = load("FETCH_SIZE"), load("WRITE_SIZE"), load("GRBM_GUI_ACTIVE")
out = {"tag": T, "workload": w, "source": "rocprofv3 --kernel-trace --pmc <group>, one run per group (tools/pmc_workload.sh)", "kernels": {}}
for k in F:
    n = len(F[k]["seen"]); fetch = F[k]["c"]["FETCH_SIZE"] * 1024 * 2; write = W.get(k, {"c": {"WRITE_SIZE": 0.0}})["c"]["WRITE_SIZE"] * 1024
    rec = {"launches": n, "kernel_ms_total_fetch_pass": F[k]["ms"], "fetch_bytes_total_x2_corrected": fetch, "write_bytes_total": write,
           "hbm_fabric_bytes_per_launch": (fetch + write) / max(n, 1), "fabric_TBps": (fetch + write) / 1e9 / max(F[k]["ms"], 1e-9)}
    if k in S and S[k]["ms"] > 0:
        c = S[k]["c"]; cyc = c["GRBM_GUI_ACTIVE"] / 8
        rec.update({"effective_clock_ghz": cyc / (S[k]["ms"] * 1e-3) / 1e9, "mfma_busy_fraction_of_simd_cycles": c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 256 * 4),
                    "wave_cycles_per_simd_cycle": c["SQ_WAVE_CYCLES"] * 4 / (cyc * 256 * 4)})
    out["kernels"][k] = rec
json.dump(out, open(f"{O}/{T}_{w}_pmc_summary.json", "w"), indent=1)
for k, r in out["kernels"].items():
    if r["kernel_ms_total_fetch_pass"] > 1.0:
        print(f"{k[:60]:60s} {r['launches']:4d} launches {r['hbm_fabric_bytes_per_launch'] / 1e9:8.2f} GB/launch {r['fabric_TBps']:5.2f} TB/s  clock {r.get('effective_clock_ghz', 0):.3f} GHz  MFMA busy {100 * r.get('mfma_busy_fraction_of_simd_cycles', 0):.1f} %")
PY
find $O -name "*_counter_collection.csv" -delete; find $O -name "*_kernel_trace.csv" -delete
