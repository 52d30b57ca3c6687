#!/bin/bash
# K1-3D against the sort interval / work block size (uniform C5 slab, rho from continuity); one gpurun call
mkdir -p gpurun_out
for si in 10 20 30; do for bp in 2048 4096; do
  python tools/bench3d.py --sort-interval $si --block-particles $bp --steps 60 --warmup 6 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('si=$si bp=$bp', 'k1=%.3f ms frac=%.4f step=%.3f ms' % (d['k1_3d_ms'], d['k1_3d_frac_of_hbm'], d['ms_per_step']), d['rho_steps'])"
done; done | tee gpurun_out/r03_sweep3d.txt
