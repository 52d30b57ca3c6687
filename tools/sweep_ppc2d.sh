#!/bin/bash
# K1 2-D against the particles per cell (1024 x 1024 cells; C2 is 64): how the kernel holds up off the benchmark's shape
mkdir -p gpurun_out
for cfg in "2048 2048 16" "1024 2048 32" "1024 1024 64" "1024 512 128" "512 512 256"; do
  set -- $cfg
  python bench.py --no-extra --no-cpu-baseline --nx $1 --ny $2 --ppc $3 --steps 40 --warmup 8 2>/dev/null | grep '^{"metric' | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('nx=$1 ny=$2 ppc=$3', 'particles=%d' % d['config']['particles_per_gpu'], 'step=%.3f ms  k1=%.3f ms frac=%.4f value=%.3e' % (d['ms_per_step'], r['kernel_ms'], r['frac'], d['value']))"
done | tee gpurun_out/r03_sweep_ppc2d.txt
