#!/bin/bash
# A/B inside one gpurun call: per-species launches of K1-3D against the one-launch multi-species form
mkdir -p gpurun_out
run() { python tools/bench3d.py --steps 40 --warmup 6 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$*', '| k1=%.3f ms frac=%.4f step=%.3f ms alive=%d charge_err=%.1e' % (d['k1_3d_ms'], d['k1_3d_frac_of_hbm'], d['ms_per_step'], d['alive'], d['charge_rel_err']))"; }
{ for rep in 1 2; do
run --no-fuse
run
run --species 2 --no-fuse
run --species 2
done; } | tee gpurun_out/r03_ab_fuse.txt
