#!/bin/bash
# Which ingredient makes a process abort in its teardown (free(): invalid pointer, after every handle was destroyed)?  Variants of the
# child of tests/test_gpu_engine.py::test_handles_left_open..., each under ab/libabortbt.so (C backtrace on SIGABRT).
out=gpurun_out/${1:-r04e}_exit_fault_probe.log
: > $out
run() {
  echo "##### $1" >> $out
  LD_PRELOAD=$PWD/ab/libabortbt.so timeout -k 5 120 python -X faulthandler -c "$2" >> $out 2>&1
  echo "exit status $?" >> $out
}
PRE="import numpy as np
b = np.random.default_rng(0).standard_normal((3000, 64)).astype(np.float32)
"
run "A: KnnEngine only, left open" "$PRE
from neighborhoodwatch_amd import KnnEngine
G = KnnEngine(0); G.set_base(b); G.search(b[:300], 5)"
run "B: ShardedKnnEngine (RCCL), closed explicitly, no torch import" "$PRE
from neighborhoodwatch_amd import ShardedKnnEngine
S = ShardedKnnEngine([0]); S.set_base(b); S.search(b[:20], 5); S.close()"
run "C: torch imported FIRST, then ShardedKnnEngine, closed explicitly" "$PRE
import torch; t = torch.ones(8, device='cuda')
from neighborhoodwatch_amd import ShardedKnnEngine
S = ShardedKnnEngine([0]); S.set_base(b); S.search(b[:20], 5); S.close()"
run "D: ShardedKnnEngine first, torch imported AFTER (the failing test's order)" "$PRE
from neighborhoodwatch_amd import ShardedKnnEngine
S = ShardedKnnEngine([0]); S.set_base(b); S.search(b[:20], 5); S.close()
import torch; t = torch.ones(8, device='cuda')"
run "E: torch only: a tensor and an RCCL-free exit" "import torch; t = torch.ones(8, device='cuda'); print(float(t.sum()))"
run "F: torch.distributed nccl group of one, destroyed (RCCL through torch, no libnwknn)" "import os, torch, torch.distributed as td
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29533')
torch.cuda.set_device(0); td.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
t = torch.ones(4, device='cuda'); td.all_reduce(t); print(float(t.sum())); td.destroy_process_group()"
run "G: NO libnwknn at all: torch's librccl.so dlopen'ed by path (RTLD_GLOBAL), then import torch" "import ctypes, importlib.util, pathlib
d = pathlib.Path(importlib.util.find_spec('torch').submodule_search_locations[0]) / 'lib'
ctypes.CDLL(str(d / 'libamdhip64.so'), mode=ctypes.RTLD_GLOBAL); ctypes.CDLL(str(d / 'librccl.so'), mode=ctypes.RTLD_GLOBAL)
import torch; t = torch.ones(8, device='cuda'); print(float(t.sum()))"
cat $out | grep -v "amdgpu.ids"
