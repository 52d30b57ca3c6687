#!/bin/bash
# C2 step time against the work-block size of K1 2-D and nearby sort intervals; one gpurun call
mkdir -p gpurun_out
for bp in 4096 8192 16384; do for si in 16 20 24; do
  python bench.py --no-extra --no-cpu-baseline --block-particles $bp --sort-interval $si --steps $((3*si)) --warmup 8 2>/dev/null | grep '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('bp=$bp si=$si', 'step=%.3f ms  k1=%.3f ms frac=%.4f value=%.3e' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['value']))"
done; done | tee gpurun_out/r03_sweep_block2d.txt
