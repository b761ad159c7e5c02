#!/bin/bash
# layout pass: one 36 KiB LDS tile rewritten in place (4 workgroups per CU) against round 2's two tiles (2 per CU), interleaved on one box; then parity
mkdir -p gpurun_out; out=gpurun_out/r04j_prep_ab.log; : > $out
for r in 1 2; do
  for lib in ab/prep_two_tiles.so neighborhoodwatch_amd/libnwknn.so; do
    echo "## round $r $lib" >> $out
    NWKNN_LIB_PATH=$PWD/$lib python tools/prep_rate.py 4 2>/dev/null >> $out
  done
done
cat $out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_engine.py tests/test_gpu_f16_exact.py tests/test_gpu_files.py -x -q -m gpu > gpurun_out/r04j_pytest.log 2>&1; echo "pytest exit status $?" >> gpurun_out/r04j_pytest.log; tail -3 gpurun_out/r04j_pytest.log
