#!/bin/bash
# round 4, first GPU call: the invariant machinery, the 1M-query C4 share, and the wall clock of the new bench legs
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_prescreen.py tests/test_gpu_f16_exact.py -x -q -m gpu -k "invariant" > gpurun_out/r04a_pytest1.log 2>&1 || { tail -30 gpurun_out/r04a_pytest1.log; exit 1; }
tail -3 gpurun_out/r04a_pytest1.log
python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k c4 --durations=5 > gpurun_out/r04a_pytest_c4.log 2>&1 || { tail -40 gpurun_out/r04a_pytest_c4.log; exit 1; }
tail -8 gpurun_out/r04a_pytest_c4.log
python bench.py --steps 2 --warmup 1 > gpurun_out/r04a_bench.json 2> gpurun_out/r04a_bench.err || { tail -30 gpurun_out/r04a_bench.err; exit 1; }
python - <<'PY'
import json
b=json.loads(open('gpurun_out/r04a_bench.json').read().strip().splitlines()[-1])
print('value',b['value'],'frac',b['roofline']['frac'],'hbm',b['roofline'].get('hbm'))
print('pre',b['exact_prescreen']['value'],b['exact_prescreen']['invariant'],b['exact_prescreen']['roofline']['frac'])
print('cpu',b['cpu_baseline']['sample'], b['cpu_baseline']['sklearn'].get('sample'))
for k,v in b['secondary'].items():
    print(k, {x:v.get(x) for x in ('value','ms_per_step','wall_s','skipped','error','invariant')}, v.get('roofline',{}).get('frac'), (v.get('exact_prescreen') or {}).get('value'))
    if k=='refcli': print(json.dumps(v.get('through_files')), json.dumps(v.get('vs_reference_published')))
print('wall', b.get('wall_s_since_process_start'))
PY
