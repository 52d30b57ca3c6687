#!/bin/bash
# K1 ablation timings (diagnostic builds, results are wrong by construction): run on the GPU box.
# usage: tools/ablate_k1.sh  -> prints kernel_ms per variant
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
for v in base:"" notail:LPA_ABLATE_NO_TAIL noatom:LPA_ABLATE_NO_ATOMICS nogather:LPA_ABLATE_NO_GATHER neither:"LPA_ABLATE_NO_ATOMICS LPA_ABLATE_NO_GATHER"; do
  name=${v%%:*}; defs=${v#*:}
  lib=$(python3 -c "from lambdapic_amd.build import build_variant; print(build_variant('$name', '$defs'.split()))") || exit 1
  LPA_LIB_PATH=$lib timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 20 --warmup 4 2>/dev/null | tail -1 | \
    python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', 'kernel_ms', round(d['roofline']['kernel_ms'],3), 'ms_per_step', round(d['ms_per_step'],3))"
done
