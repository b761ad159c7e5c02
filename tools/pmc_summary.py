"""Summarises the rocprofv3 passes of tools/profile_round.sh into the files kept under profiles/.
usage: pmc_summary.py <dir with the passes> <tag>   ->  <dir>/<tag>_c3_pmc_summary.json, traffic_c3.json, traffic_c3_prescreen.json"""
import collections
import csv
import glob
import json
import sys

O, tag = sys.argv[1], sys.argv[2]


def passes(prefix):
    out = {}
    for d in glob.glob(f"{O}/{prefix}_*/"):
        name = d.rstrip("/").split(prefix + "_")[-1]
        files = glob.glob(d + "*/*_counter_collection.csv")
        if files:
            out[name] = files[0]
    return out


def collect(csv_file, want):
    """-> {counter: sum}, kernel ms (each dispatch once), launches, kernel name  for rows whose kernel name satisfies want()."""
    acc = collections.defaultdict(float); ms = 0.0; seen = set(); name = None
    for r in csv.DictReader(open(csv_file)):
        if not want(r["Kernel_Name"]):
            continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"])
        key = r.get("Dispatch_Id") or r["Start_Timestamp"]
        if key not in seen:
            seen.add(key); ms += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6; name = r["Kernel_Name"]
    return acc, ms, len(seen), name


def traffic(ps, want):
    f, fms, fn, name = collect(ps["FETCH_SIZE"], want)
    w, _, _, _ = collect(ps["WRITE_SIZE"], want)
    fetch = f["FETCH_SIZE"] * 1024 * 2           # KB -> B, x2: gfx950 tallies 128-B requests at 64 B (MI355X_MICROARCH.md, HBM)
    write = w["WRITE_SIZE"] * 1024
    return {"kernel": name, "launches": fn, "fetch_bytes_total_x2_corrected": fetch, "write_bytes_total": write,
            "hbm_fabric_bytes_per_launch": (fetch + write) / max(fn, 1), "kernel_ms_total_fetch_pass": fms,
            "fabric_TBps": (fetch + write) / 1e9 / max(fms, 1e-9)}


def sq(ps, want):
    a, ms, n, name = collect(ps["GRBM_GUI_ACTIVE"], want)
    cyc = a["GRBM_GUI_ACTIVE"] / 8                       # summed over the 8 XCDs
    return {"kernel": name, "launches": n, "kernel_ms_total": ms, "effective_clock_ghz": cyc / (ms * 1e-3) / 1e9,
            "mfma_busy_fraction_of_simd_cycles": a["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 256 * 4),
            "wave_cycles_per_simd_cycle": a["SQ_WAVE_CYCLES"] * 4 / (cyc * 256 * 4), "counters": dict(a)}


direct = lambda k: "nwk_tile_kernel<" in k and "false" in k                  # the filtered launches of the fp32 tile kernel
pre_tile = lambda k: "nwk_tile_kernel_f16" in k
rescore = lambda k: "nwk_rescore_kernel" in k
src = "rocprofv3 --kernel-trace --pmc <counters>, one pass per counter group (FETCH_SIZE | WRITE_SIZE | GRBM_GUI_ACTIVE SQ_*), tools/profile_round.sh"

pd, pp = passes("pmc_direct"), passes("pmc_pre")
summary = {"tag": tag, "source": src}
if pd:
    t = traffic(pd, direct); s = sq(pd, direct)
    summary["direct"] = {"traffic": t, "sq": s}
    json.dump({"workload": "c3", "kernel": t["kernel"], "launches": t["launches"], "hbm_fabric_bytes_per_launch": t["hbm_fabric_bytes_per_launch"],
               "fetch_bytes_total_x2_corrected": t["fetch_bytes_total_x2_corrected"], "write_bytes_total": t["write_bytes_total"],
               "kernel_ms_total": t["kernel_ms_total_fetch_pass"], "fabric_TBps": t["fabric_TBps"], "effective_clock_ghz": s["effective_clock_ghz"],
               "mfma_busy_fraction": s["mfma_busy_fraction_of_simd_cycles"], "source": src + f", {tag}"}, open(f"{O}/traffic_c3.json", "w"), indent=1)
if pp:
    tt = traffic(pp, pre_tile); tr = traffic(pp, rescore); s = sq(pp, pre_tile)
    summary["prescreen"] = {"tile_traffic": tt, "rescore_traffic": tr, "tile_sq": s}
    json.dump({"workload": "c3", "path": "exact search through the fp16 pre-screen (option prescreen=1)", "tile": tt, "rescore": tr,
               "tile_effective_clock_ghz": s["effective_clock_ghz"], "tile_mfma_busy_fraction": s["mfma_busy_fraction_of_simd_cycles"],
               "source": src + f", {tag}"}, open(f"{O}/traffic_c3_prescreen.json", "w"), indent=1)
json.dump(summary, open(f"{O}/{tag}_c3_pmc_summary.json", "w"), indent=1)
for key in ("direct", "prescreen"):
    if key in summary:
        v = summary[key]
        t = v.get("traffic") or v.get("tile_traffic"); s = v.get("sq") or v.get("tile_sq")
        print(f"{key}: {t['kernel']}: {t['launches']} launches, {t['hbm_fabric_bytes_per_launch'] / 1e9:.1f} GB fabric per launch ({t['fabric_TBps']:.2f} TB/s), "
              f"clock {s['effective_clock_ghz']:.3f} GHz, MFMA busy {s['mfma_busy_fraction_of_simd_cycles'] * 100:.1f} % of SIMD cycles")
