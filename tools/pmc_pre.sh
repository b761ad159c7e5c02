#!/bin/bash
# HBM/fabric traffic of the pre-screen tile kernel and of the rescore kernel on one workload (two PMC passes: FETCH_SIZE, WRITE_SIZE)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
w=${1:-c3}
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmcpre_${w}_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmcpre_${w}_$c -- python3 bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline --opt prescreen=1 > gpurun_out/pmcpre_${w}_$c.json 2> gpurun_out/pmcpre_${w}_$c.err || exit 1
done
python3 - <<PY
import csv, glob, json
out = {"workload": "$w", "path": "exact search through the fp16 pre-screen (option prescreen=1)"}
for kern, tag in (("nwk_tile_kernel_f16", "tile"), ("nwk_rescore_kernel", "rescore")):
    rec = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob(f"gpurun_out/pmcpre_${w}_{c}/*/*_counter_collection.csv")[0]
        tot = 0.0; ms = 0.0; n = 0; name = None
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"] and r["Counter_Name"] == c:
                tot += float(r["Counter_Value"]); ms += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6; n += 1; name = r["Kernel_Name"]
        rec[c] = (tot, ms, n, name)
    fetch = rec["FETCH_SIZE"][0] * 1024 * 2          # KB -> B, x2: gfx950 tallies 128-B requests at 64 B (MI355X_MICROARCH.md)
    write = rec["WRITE_SIZE"][0] * 1024
    n = rec["FETCH_SIZE"][2]
    out[tag] = {"kernel": rec["FETCH_SIZE"][3], "launches": n, "fetch_bytes_total_x2_corrected": fetch, "write_bytes_total": write,
                "hbm_fabric_bytes_per_launch": (fetch + write) / max(n, 1), "kernel_ms_total_fetch_pass": rec["FETCH_SIZE"][1],
                "fabric_TBps": (fetch + write) / 1e9 / max(rec["FETCH_SIZE"][1], 1e-9)}
out["source"] = "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes (tools/pmc_pre.sh)"
json.dump(out, open(f"gpurun_out/traffic_${w}_prescreen.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
