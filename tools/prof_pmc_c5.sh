#!/bin/bash
# PMC passes for the C5-slab leg (tools/bench_c5leg.py): HBM traffic of the two-species K1-3D launch; same rules as
# prof_pmc.sh (counters in their own runs, --kernel-trace only)
set -u
OUT=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
pass() {
  name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$ROOT/$OUT/$name" -- \
     python3 "$ROOT/tools/bench_c5leg.py" 6 12 > "$ROOT/$OUT/$name.log" 2>&1
  echo "pass $name exit $?"
}
pass sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass fetch FETCH_SIZE GRBM_GUI_ACTIVE
pass write WRITE_SIZE
