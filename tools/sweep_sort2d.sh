#!/bin/bash
# C2 step time against the sort interval (rho from continuity: one real deposit per sort); one gpurun call
mkdir -p gpurun_out
for si in 20 30 40 60; do
  python bench.py --no-extra --no-cpu-baseline --sort-interval $si --steps $((2*si)) --warmup 8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('si=$si', 'step=%.3f ms  k1=%.3f ms frac=%.4f value=%.3e' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['value']), d['config']['rho_steps'])"
done | tee gpurun_out/r03_sweep_sort2d.txt
