#!/bin/bash
# hot plasmas with the sort interval following the overflow list (the default) against the fixed interval, 160 timed steps
# after 8 of warm-up: the controller needs two or three sorts to settle
mkdir -p gpurun_out
for u in 0.1 0.2 0.5 1.0; do
  python bench.py --no-extra --no-cpu-baseline --uth $u --steps 160 --warmup 8 2>/dev/null | grep '^{"metric' | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('2-D uth=$u', 'step=%.3f ms  k1(tiled)=%.3f ms value=%.3e' % (d['ms_per_step'], r['kernel_ms'], d['value']), d['config']['rho_steps'])"
done | tee gpurun_out/r03_sweep_hot_adaptive.txt
for u in 0.2 0.5 1.0; do for f in "" "--fixed-sort"; do
  python tools/bench3d.py --uth $u --steps 100 $f 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('3-D uth=$u $f', 'k1(tiled)=%.3f ms step=%.3f ms' % (d['k1_3d_ms'], d['ms_per_step']), d['rho_steps'])"
done; done | tee -a gpurun_out/r03_sweep_hot_adaptive.txt
