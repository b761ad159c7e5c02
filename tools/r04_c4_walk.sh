#!/bin/bash
# C4 per-GPU share (1M queries): super-tile shapes at the real query count, one timed search each (unprofiled arms, same box, interleaved twice)
out=gpurun_out/r04p_c4gpu_walk.log; : > $out
for r in 1 2; do
  for cfg in 4:0 5:0 3:0 6:0; do
    sh=${cfg%%:*}; wq=${cfg##*:}
    python bench.py --workload c4gpu --steps 1 --warmup 0 --no-cpu-baseline --opt super_q_shift=$sh,super_walk_q=$wq 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('round $r  %2d q-tiles x %2d b-strips, walk_q $wq: %.0f q/s, %.2f s/step, tile %.1f TF = %.3f' % (1<<$sh, 64>>$sh, b['value'], b['ms_per_step']/1e3, b['roofline']['achieved'], b['roofline']['frac']))" >> $out
    tail -1 $out
  done
done
