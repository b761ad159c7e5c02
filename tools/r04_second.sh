#!/bin/bash
# round 4, second GPU call: fp32 kernel with one per-lane DMA offset + no stamp code (A/B vs the committed build), parity, the refcli file leg
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_engine.py tests/test_gpu_prescreen.py tests/test_gpu_f16_exact.py tests/test_gpu_distributed_rccl.py -x -q -m gpu > gpurun_out/r04b_pytest.log 2>&1 || { tail -30 gpurun_out/r04b_pytest.log; exit 1; }
tail -2 gpurun_out/r04b_pytest.log
bash tools/ab_direct.sh "10000000" ab/r04_base.so neighborhoodwatch_amd/libnwknn.so 2 > gpurun_out/r04b_ab_f32.log 2>&1
cat gpurun_out/r04b_ab_f32.log
bash tools/ab_direct.sh "1250000" ab/r04_base.so neighborhoodwatch_amd/libnwknn.so 2 >> gpurun_out/r04b_ab_f32.log 2>&1
tail -4 gpurun_out/r04b_ab_f32.log
python bench.py --workload refcli --files --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r04b_refcli.json 2> gpurun_out/r04b_refcli.err || { tail -20 gpurun_out/r04b_refcli.err; exit 1; }
python - <<'PY'
import json
b=json.loads(open('gpurun_out/r04b_refcli.json').read().strip().splitlines()[-1])
print(b['value'], b['roofline']['frac'], json.dumps(b['through_files']))
PY
