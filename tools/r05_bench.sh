#!/bin/bash
# round 5: smoke + the bench line as the driver runs it (20 steps, 5 warm-ups), logs under gpurun_out/<tag>_* (the GPU suite runs in its own call: profiles/r05t_pytest_gpu.log)
set -o pipefail
tag=${1:-r05v}
mkdir -p gpurun_out
true; rc=0


[ $rc -ne 0 ] && exit $rc
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${tag}_smoke.log 2>&1; rc=$?
echo "smoke exit status $rc" >> gpurun_out/${tag}_smoke.log; tail -2 gpurun_out/${tag}_smoke.log
[ $rc -ne 0 ] && exit $rc
SECONDS=0
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${tag}_c3_bench.json 2> gpurun_out/${tag}_c3_bench.err; rc=$?
echo "bench.py --gpus 1 --steps 20 --warmup 5: ${SECONDS} s of wall clock, exit status $rc" | tee gpurun_out/${tag}_c3_bench.wall
[ $rc -ne 0 ] && { tail -30 gpurun_out/${tag}_c3_bench.err; exit $rc; }
python - <<PY
import json
b=json.loads(open('gpurun_out/${tag}_c3_bench.json').read().strip().splitlines()[-1])
rf=b['roofline']
print('value',b['value'],'frac',rf['frac'],'bare',rf.get('bare_loop_tflops'),rf.get('bare_loop_clock_ghz'),rf.get('frac_of_bare_loop'),'hbm',rf['hbm']['achieved'])
p=b['exact_prescreen']; print('pre',p['value'],p['invariant'],p['roofline']['frac'],p['roofline'].get('bare_loop_tflops'),p['roofline'].get('frac_of_bare_loop'))
print('cpu',b['cpu_baseline']['value'], b['cpu_baseline']['sklearn'].get('value'))
for k,v in b['secondary'].items():
    print(k, {x:v.get(x) for x in ('value','ms_per_step','wall_s','skipped','error')}, v.get('roofline',{}).get('frac'), v.get('roofline',{}).get('frac_of_bare_loop'), (v.get('exact_prescreen') or {}).get('value'))
    if k=='refcli': print(json.dumps(v.get('through_files')))
print('wall', b.get('wall_s_since_process_start'))
PY
