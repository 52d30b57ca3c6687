#!/bin/bash
# C2's plasma as a flow along x (mean u_x = 0 ... 10): every particle changes cell every other step, all the same way
for d in 0 0.3 1 3 10; do
  echo "== drift u_x $d"
  timeout -k 10 200 python bench.py --drift $d --no-extra --no-cpu-baseline --steps 80 --warmup 10 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('ms/step %.3f  K1 %.3f ms  value %.3e' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['value']))
" || exit 1
done
