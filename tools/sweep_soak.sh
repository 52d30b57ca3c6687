#!/bin/bash
# laser-target configs deep into the hot phase: the sort-interval controller's threshold against fixed intervals
mkdir -p gpurun_out
for f in 0.003 0 0.01 0.03 0.1; do
  echo "== overflow_sort_fraction $f"
  timeout -k 10 300 python tools/soak.py --fraction $f 2>&1 | grep -E "C3 step +(2500|3000)|C5 slab step +(320|360|400)|ok|Error|error" || exit 1
done
