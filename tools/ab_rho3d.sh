#!/bin/bash
# A/B inside one gpurun call: rho deposited in every step against rho from the continuity equation (3-D slab of C5)
set -e
mkdir -p gpurun_out
for mode in deposited continuity deposited continuity; do
  python tools/bench3d.py --rho $mode --steps 40 --warmup 6
done > gpurun_out/r03_ab_rho3d.txt 2>&1
cat gpurun_out/r03_ab_rho3d.txt
