#!/bin/bash
# round 4: the parity suites with every fp16 tile kernel form forced on every shape (the invariant hook asserts after every search), MaxSim soak
mkdir -p gpurun_out; out=gpurun_out/r04n_forced_forms_parity.log; touch $out
for form in ${FORMS:-4 3 1}; do
  echo "##### NWKNN_DEFAULT_OPTS=f16_mfma=$form" >> $out
  NWKNN_DEFAULT_OPTS=f16_mfma=$form python -m pytest tests/test_gpu_prescreen.py tests/test_gpu_f16_exact.py tests/test_gpu_prescreen_fuzz.py tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider 2>&1 | tail -6 >> $out
  echo "exit status ${PIPESTATUS[0]}" >> $out
  tail -2 $out
done
true
true
