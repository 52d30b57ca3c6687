#!/bin/bash
# same-box sweep of bench.py runtime parameters: tools/sweep_bench.sh "flags" "flags" ...
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
for f in "$@"; do
  timeout -k 10 300 python3 bench.py --no-cpu-baseline $f 2>/dev/null | tail -1 | \
    python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f', 'kernel_ms', round(d['roofline']['kernel_ms'],3), 'ms_per_step', round(d['ms_per_step'],3))"
done
