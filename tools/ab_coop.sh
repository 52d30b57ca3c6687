#!/bin/bash
# A/B inside one gpurun call: cooperative deposit on the padded order against the product, several sort intervals
for si in "$@"; do
  for fl in "--order padded" ""; do
    echo -n "== sort_interval=$si $fl: "
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-extra --steps 40 --warmup 8 --sort-interval $si $fl 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', round(d['ms_per_step'],3), 'kernel_ms', round(d['roofline']['kernel_ms'],3), 'alive', d['config']['alive_rank0'])"
  done
done
