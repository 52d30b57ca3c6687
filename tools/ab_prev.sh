#!/bin/bash
# same-box A/B of the product library against a previously built csrc/build/liblambdapic_amd_prev.so
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
for rep in 1 2; do
  for lib in lambdapic_amd/csrc/build/liblambdapic_amd_prev.so lambdapic_amd/liblambdapic_amd.so; do
    LPA_LIB_PATH=$ROOT/$lib timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 40 --warmup 8 "$@" 2>/dev/null | tail -1 | \
      python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', 'kernel_ms', round(d['roofline']['kernel_ms'],3), 'ms_per_step', round(d['ms_per_step'],3))"
  done
done
