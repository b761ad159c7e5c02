#!/bin/bash
# SQ / LDS counters of the fp16 tile kernel forms side by side: tools/f16_forms.py under rocprofv3 --pmc, two passes.
# usage (GPU box, repo root): tools/pmc_forms.sh TAG "<forms>" d B Q
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; forms=${2:-3,4}; d=${3:-1024}; B=${4:-4000000}; Q=${5:-32768}
O=gpurun_out/pmcforms_$tag
rm -rf $O; mkdir -p $O
rocprofv3 -L > $O/counters_available.txt 2>&1 || true
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES --output-format csv -d $O/p1 -- python3 tools/f16_forms.py $forms $d $B $Q 1 > $O/p1.log 2>&1 || { tail -5 $O/p1.log; exit 1; }
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_F16 --output-format csv -d $O/p2 -- python3 tools/f16_forms.py $forms $d $B $Q 1 > $O/p2.log 2>&1 || { tail -5 $O/p2.log; }
python3 - <<PY
import csv, glob, json, collections
out = {}
for ps in ("p1", "p2"):
    fs = glob.glob("$O/%s/*/*_counter_collection.csv" % ps)
    if not fs: continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); ms = collections.defaultdict(float); seen = set()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        if "nwk_tile_kernel_f16" not in k or "true" in k: continue
        k = k.split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r.get("Dispatch_Id") or r["Start_Timestamp"])
        if key not in seen:
            seen.add(key); ms[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    for k in acc:
        a = acc[k]; cyc = a["GRBM_GUI_ACTIVE"] / 8
        e = out.setdefault(k, {})
        e[ps + "_kernel_ms"] = ms[k]; e[ps + "_clock_ghz"] = cyc / (ms[k] * 1e-3) / 1e9
        simd_cycles = cyc * 256 * 4
        for c, v in a.items():
            if c == "GRBM_GUI_ACTIVE": continue
            e[c] = v
            e[c + "_per_simd_cycle"] = v / simd_cycles
json.dump(out, open("$O/summary.json", "w"), indent=1)
for k, e in out.items():
    print(k)
    for c in sorted(e):
        if c.endswith("_per_simd_cycle") or c.endswith("ghz") or c.endswith("_ms"): print("   %-48s %.4f" % (c, e[c]))
PY
