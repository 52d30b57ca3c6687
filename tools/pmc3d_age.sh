#!/bin/bash
# LDS counters of K1-3D right after a sort (interval 1) and aged (interval 10)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc3d_age
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for si in 1 10; do
  timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES \
     --kernel-trace --output-format csv -d $OUT/si$si -- python3 $ROOT/tools/bench3d.py --sort-interval $si --steps 10 --warmup 2 > $OUT/si$si.log 2>&1
  echo "== sort interval $si (exit $?)"
  python3 $ROOT/tools/pmc_summary.py $OUT/si$si k_push_deposit_tiled_3d
done
