#!/bin/bash
for bp in 6144 8192 12288 16384 24576; do
    echo -n "== block_particles=$bp: "
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-extra --steps 40 --warmup 8 --block-particles $bp 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', round(d['ms_per_step'],3), 'kernel_ms', round(d['roofline']['kernel_ms'],3))"
done
