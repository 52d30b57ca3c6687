#!/bin/bash
# K1-3D experiment timings for compile-time variants: tools/exp_k13.sh "name:DEF1 DEF2[:bench3d flags]" ...
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
for v in "$@"; do
  name=${v%%:*}; rest=${v#*:}; defs=${rest%%:*}; flags=""
  case "$rest" in *:*) flags=${rest#*:};; esac
  lib=$(python3 -c "from lambdapic_amd.build import build_variant; print(build_variant('$name', '$defs'.split()))" 2>/dev/null) || exit 1
  LPA_LIB_PATH=$lib timeout -k 10 200 python3 tools/bench3d.py $flags 2>/dev/null | tail -1 | \
    python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', '$flags', 'ms_per_step', round(d['ms_per_step'],3), 'p/s', '%.3g' % d['value'])"
done
