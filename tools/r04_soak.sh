#!/bin/bash
# round 4: MaxSim tests with the invariant, a fuzz soak at HEAD with the invariant hook on every search, the C5 bench line
mkdir -p gpurun_out
python -m pytest tests/test_maxsim.py -x -q -m gpu > gpurun_out/r04i_pytest_maxsim.log 2>&1; echo "pytest exit status $?" >> gpurun_out/r04i_pytest_maxsim.log
tail -3 gpurun_out/r04i_pytest_maxsim.log
NWKNN_FUZZ_SEED0=40000 NWKNN_FUZZ_CASES=${1:-4000} python -m pytest tests/test_gpu_prescreen_fuzz.py -x -q -m gpu -p no:cacheprovider > gpurun_out/r04i_fuzz_soak_invariant.log 2>&1; echo "pytest exit status $?" >> gpurun_out/r04i_fuzz_soak_invariant.log
tail -3 gpurun_out/r04i_fuzz_soak_invariant.log
python bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r04i_c5_bench.json 2> gpurun_out/r04i_c5_bench.err; echo "c5 exit $?"
python - <<'PY'
import json
b=json.loads(open('gpurun_out/r04i_c5_bench.json').read().strip().splitlines()[-1])
p=b['exact_prescreen']; print('c5 direct', b['value'], 'pre', p['value'], p['identical_to_direct'], p['invariant'], p['roofline']['frac'], p['roofline'].get('frac_of_bare_loop'), p['breakdown_ms_per_step'])
PY
