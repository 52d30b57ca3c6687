#!/bin/bash
# A/B of the in-tile order of the 3-D store: LPA_ORDER_STRIPED (product) against LPA_ORDER_COLUMN, on the uniform slab
# (tools/bench3d.py) and on the C5 slab leg (tools/bench_c5leg.py).
set -e
mkdir -p gpurun_out/ab_column
cd "$(dirname "$0")/.."
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "cell_sort_properties" > gpurun_out/ab_column/sort_test.log 2>&1
for o in striped column; do
  python tools/bench3d.py --order $o > gpurun_out/ab_column/bench3d_$o.json
  python tools/bench3d.py --order $o --sort-interval 5 > gpurun_out/ab_column/bench3d_${o}_s5.json
  python tools/bench3d.py --order $o --sort-interval 20 > gpurun_out/ab_column/bench3d_${o}_s20.json
  python tools/bench_c5leg.py 40 12 $o > gpurun_out/ab_column/c5leg_$o.json
done
tail -2 gpurun_out/ab_column/sort_test.log
for f in gpurun_out/ab_column/*.json; do echo $f; python - "$f" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d[k] for k in ("k1_3d_ms", "ms_per_step", "charge_rel_err", "step_ms", "roofline", "value") if k in d})
PY
done
