#!/bin/bash
# hot plasmas against the sort interval (C2's box): where does the time go when particles outrun the tile margin?
for u in 0.2 0.5 1.0; do for si in 20 10 5 3 2; do
  python bench.py --no-extra --no-cpu-baseline --uth $u --sort-interval $si --steps $((si*4 > 40 ? si*4 : 40)) --warmup 8 2>/dev/null | grep '^{"metric' | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('uth=$u sort_interval=$si', 'step=%.3f ms  k1(tiled)=%.3f ms value=%.3e' % (d['ms_per_step'], r['kernel_ms'], d['value']))"
done; done | tee gpurun_out/r03_sweep_hot2d.txt
