#!/bin/bash
# PMC passes for the bench workload (run on the GPU box through gpurun).  Counters are collected in
# their own runs with --kernel-trace only (never combined with sys/hip/hsa tracing).
# usage: tools/prof_pmc.sh <outdir> [bench args...]
set -u
OUT=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
pass() {
  name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$ROOT/$OUT/$name" -- \
     python3 "$ROOT/bench.py" --no-cpu-baseline --no-extra $BENCH_ARGS > "$ROOT/$OUT/$name.log" 2>&1
  echo "pass $name exit $?"
}
BENCH_ARGS="${BENCH_ARGS:---steps 4 --warmup 2}"
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU
pass sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass fetch FETCH_SIZE GRBM_GUI_ACTIVE
pass write WRITE_SIZE
pass atom TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum
