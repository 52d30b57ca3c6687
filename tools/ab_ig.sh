#!/bin/bash
# C2 with inv_gamma recomputed in the fused kernels (product) against streamed like the reference's kernel; one gpurun call
for rep in 1 2 3; do for m in streamed recomputed; do
  python bench.py --no-extra --no-cpu-baseline --steps 60 --warmup 10 --inv-gamma $m 2>/dev/null | grep '^{"metric' | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('%-10s' % '$m', 'C2 k1=%.3f ms frac=%.4f step=%.3f ms value=%.3e' % (r.get('kernel_ms', 0), r['frac'], d['ms_per_step'], d['value']))"
done; done
