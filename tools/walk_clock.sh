#!/bin/bash
# Super-tile walk orders of the fp16 tile kernel judged on CLOCK + WALL + FABRIC BYTES together, every arm under the profiler
# (MI355X_MICROARCH.md: never compare a profiled arm with an unprofiled one).  Per arm two rocprofv3 passes of the same bench command:
#   pass 1  --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES   -> effective clock, MFMA busy, kernel ms
#   pass 2  --pmc FETCH_SIZE                                                -> bytes from beyond L2 (x2: gfx950 tallies 128-B requests at 64 B)
# usage: tools/walk_clock.sh <workload> "<bench args>" "<shift:walkq ...>" <out.log>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
w=${1:-c4slice}; args=${2:---steps 2 --warmup 1}; cfgs=${3:-4:0 2:1 3:0}; out=${4:-gpurun_out/walk_clock_$w.log}
: > $out
for cfg in $cfgs; do
  sh=${cfg%%:*}; wq=${cfg##*:}
  for pass in sq fetch; do
    ctr="GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES"; [ $pass = fetch ] && ctr="FETCH_SIZE"
    rm -rf gpurun_out/wc_${pass}
    rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d gpurun_out/wc_${pass} -- python3 bench.py --workload $w $args --no-cpu-baseline --no-secondary --opt super_q_shift=$sh,super_walk_q=$wq${EXTRA_OPT} > gpurun_out/wc_${pass}.json 2> gpurun_out/wc_${pass}.err || { tail -5 gpurun_out/wc_${pass}.err; exit 1; }
  done
  python3 - >> $out <<PY
import collections, csv, glob, json
def collect(d):
    f = glob.glob(f"gpurun_out/{d}/*/*_counter_collection.csv")[0]
    acc = collections.defaultdict(float); ms = 0.0; seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "nwk_tile_kernel_f16x" not in k: continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"])
        key = r.get("Dispatch_Id") or r["Start_Timestamp"]
        if key not in seen:
            seen.add(key); ms += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    return acc, ms, len(seen)
a, ms, n = collect("wc_sq"); f, fms, fn = collect("wc_fetch")
cyc = a["GRBM_GUI_ACTIVE"] / 8
b = json.loads(open("gpurun_out/wc_sq.json").read().strip().splitlines()[-1])
rf = b["roofline"]          # the f16 workloads, or an fp32 workload run with EXTRA_OPT=,prescreen=1: the main leg IS the fp16 kernel's
print(f"$w {1<<$sh:2d} q-tiles x {64>>$sh:2d} b-strips, {'query' if $wq else 'base'} tile fastest: f16x launches {n}, kernel {ms:9.2f} ms (fetch pass {fms:9.2f}), "
      f"clock {cyc/(ms*1e-3)/1e9:.3f} GHz, MFMA busy {a['SQ_VALU_MFMA_BUSY_CYCLES']/(cyc*1024)*100:.1f} %, busy x clock / 2.4 = {a['SQ_VALU_MFMA_BUSY_CYCLES']/(cyc*1024)*cyc/(ms*1e-3)/2.4e9:.3f}, "
      f"fabric {f['FETCH_SIZE']*2048/1e9/max(fn,1):7.1f} GB per launch, bench: {b['value']:.0f} q/s, tile {rf['achieved']:.1f} TF = {rf['achieved']/2500.0:.3f} of 2.5 PF")
PY
  tail -1 $out
done
rm -rf gpurun_out/wc_sq gpurun_out/wc_fetch
