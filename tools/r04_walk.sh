#!/bin/bash
# round 4: bare-loop calibration per waves/SIMD, walk orders on clock + wall + bytes, the query-sharded per-GPU piece
mkdir -p gpurun_out
NWKNN_CALIB_VERBOSE=1 python - > gpurun_out/r04h_calibration.log 2>&1 <<'PY'
from neighborhoodwatch_amd import KnnEngine
with KnnEngine(0) as e:
    for r in range(2):
        print("f16", e.mfma_ceiling("f16", 2.0), flush=True)
        print("f32", e.mfma_ceiling("f32", 3.0), flush=True)
PY
cat gpurun_out/r04h_calibration.log
bash tools/walk_clock.sh c4slice "--steps 2 --warmup 1" "4:0 2:1 3:0 4:1" gpurun_out/r04h_walk_clock_c4slice.log
EXTRA_OPT=",prescreen=1" bash tools/walk_clock.sh c3 "--steps 2 --warmup 1" "4:0 2:1 3:0" gpurun_out/r04h_walk_clock_c3pre.log
python bench.py --workload c3qshard --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r04h_c3qshard.json 2> gpurun_out/r04h_c3qshard.err
python - <<'PY'
import json
b=json.loads(open('gpurun_out/r04h_c3qshard.json').read().strip().splitlines()[-1])
print('c3qshard direct', b['value'], b['ms_per_step'], 'pre', b['exact_prescreen']['value'], b['exact_prescreen']['ms_per_step'], b['exact_prescreen']['breakdown_ms_per_step'])
PY
