#!/bin/bash
# K1 2-D (C2's box and particle count) against the thermal spread: hotter plasmas change cell more often (more crossers in the
# second pass, faster decay of the order between sorts) and, above ~0.3, send particles to the overflow list (more than a
# tile margin per sort interval)
mkdir -p gpurun_out
for u in 0.01 0.0442 0.1 0.2 0.5 1.0; do
  python bench.py --no-extra --no-cpu-baseline --uth $u --steps 40 --warmup 8 2>/dev/null | grep '^{"metric' | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('uth=$u', 'step=%.3f ms  k1=%.3f ms frac=%.4f value=%.3e' % (d['ms_per_step'], r['kernel_ms'], r['frac'], d['value']))"
done | tee gpurun_out/r03_sweep_uth2d.txt
for p in 4 8 16 32 64; do
  python tools/bench3d.py --ppc $p --nx $((64*8/p > 8 ? 64*8/p : 8)) 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('3-D ppc=$p cells', d['cells'], 'particles', d['particles'], 'k1=%.3f ms frac=%.4f step=%.3f ms overflow %s' % (d['k1_3d_ms'], d['k1_3d_frac_of_hbm'], d['ms_per_step'], d['overflow_last_step']))"
done | tee gpurun_out/r03_sweep_ppc3d.txt
