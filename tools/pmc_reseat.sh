#!/bin/bash
# LDS counters of K1 with and without the in-kernel cell-index sort (run through gpurun)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_reseat
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mode in reseat noreseat; do
  fl=""; [ $mode = reseat ] && fl="--reseat"
  timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR \
     --kernel-trace --output-format csv -d $OUT/$mode -- python3 $ROOT/bench.py --no-cpu-baseline --no-extra --steps 12 --warmup 4 $fl > $OUT/$mode.log 2>&1
  echo "== $mode (exit $?)"
  python3 $ROOT/tools/pmc_summary.py $OUT/$mode k_push_deposit_tiled_2d
done
