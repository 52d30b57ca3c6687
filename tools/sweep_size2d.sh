#!/bin/bash
# C2's plasma at other sizes (64 per cell): from launch-bound boxes to 268 M particles
mkdir -p gpurun_out
for cfg in "128 128" "256 256" "512 512" "1024 1024" "2048 1024" "2048 2048"; do
  set -- $cfg
  python bench.py --no-extra --no-cpu-baseline --nx $1 --ny $2 --ppc 64 --steps 40 --warmup 8 2>/dev/null | grep '^{"metric' | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('nx=$1 ny=$2', 'particles=%d' % d['config']['particles_per_gpu'], 'step=%.3f ms  k1=%.3f ms frac=%.4f value=%.3e' % (d['ms_per_step'], r['kernel_ms'], r['frac'], d['value']))"
done | tee gpurun_out/r03_sweep_size2d.txt
