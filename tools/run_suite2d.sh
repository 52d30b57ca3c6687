#!/bin/bash
# the 2-D part of the -m gpu suite + the C2 bench line (quick check after a change of the 2-D kernels)
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu -k "not 3d and not engine3d" > gpurun_out/suite2d.log 2>&1
rc=$?
tail -5 gpurun_out/suite2d.log
[ $rc -ne 0 ] && exit $rc
python bench.py --no-extra --no-cpu-baseline --steps 60 --warmup 10 2>/dev/null | grep '^{"metric' | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('C2 k1=%.3f ms frac=%.4f step=%.3f ms value=%.3e' % (r.get('kernel_ms', 0), r['frac'], d['ms_per_step'], d['value']))"
