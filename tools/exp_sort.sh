#!/bin/bash
# sort experiment: per-variant k_scatter_tiled time from rocprofv3 on bench.py; tools/exp_sort.sh "name:DEFS" ...
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
for v in "$@"; do
  name=${v%%:*}; defs=${v#*:}
  lib=$(python3 -c "from lambdapic_amd.build import build_variant; print(build_variant('$name', '$defs'.split()))" 2>/dev/null) || exit 1
  out=$ROOT/gpurun_out/exps_$name
  (cd /tmp && TMPDIR=/tmp LPA_LIB_PATH=$lib timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $ROOT/bench.py --no-cpu-baseline > $out.log 2>&1)
  f=$(ls $out/*/*kernel_stats.csv | head -1)
  python3 -c "
import csv
for r in csv.DictReader(open('$f')):
    if 'k_scatter_tiled' in r['Name'] or 'k_cell_count_tiled' in r['Name']: print('$name', r['Name'][:22], r['Calls'], 'max_ms', round(float(r['MaxNs'])/1e6,3))
"
done
