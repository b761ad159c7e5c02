#!/bin/bash
# (no GPU) the CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer over the CPU tests that drive it (golden vectors, formats, MaxSim
# oracle, host logic).  GPU sanitizers are not available on this pool; the oracle is the checker every parity claim rests on, so it gets the pass.
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
gcc -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -fopenmp -ffp-contract=off -shared -fPIC $root/oracle/nwknn_oracle.c -o /tmp/libnwknn_oracle_asan.so -lm
cat > /tmp/asan_run.py <<PY
import sys, pathlib
sys.path.insert(0, "$root")
from oracle import cpu_oracle
cpu_oracle._LIB_PATH = pathlib.Path("/tmp/libnwknn_oracle_asan.so")
cpu_oracle.build = lambda force=False: cpu_oracle._LIB_PATH
import pytest
sys.exit(pytest.main(["-x", "-q", "-m", "not gpu", "tests/test_oracle_golden.py", "tests/test_formats.py", "tests/test_maxsim.py", "tests/test_host_logic.py", "-p", "no:cacheprovider"]))
PY
cd $root && LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 OMP_NUM_THREADS=4 python /tmp/asan_run.py
