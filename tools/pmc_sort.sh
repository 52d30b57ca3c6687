#!/bin/bash
# HBM traffic of the re-sort kernels (k_scatter_tiled, k_cell_count_tiled) on the bench workload; counters in their own
# passes with --kernel-trace only.  usage: tools/pmc_sort.sh <outdir>
set -u
OUT=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$ROOT/$OUT/$c" -- \
     python3 "$ROOT/bench.py" --no-cpu-baseline --no-extra --steps 42 --warmup 2 > "$ROOT/$OUT/$c.log" 2>&1 || exit 1
  python3 "$ROOT/tools/pmc_summary.py" "$ROOT/$OUT/$c" k_scatter
  python3 "$ROOT/tools/pmc_summary.py" "$ROOT/$OUT/$c" k_cell_count
  rm -rf "$ROOT/$OUT/$c"
done
