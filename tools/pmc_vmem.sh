#!/bin/bash
# VMEM / LDS queue counters of K1 2-D for alternative builds (diagnostic): bash tools/pmc_vmem.sh name1 name2 ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_vmem
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for lib in product "$@"; do
  if [ $lib = product ]; then unset LPA_LIB_PATH; else export LPA_LIB_PATH=$ROOT/lambdapic_amd/csrc/build/liblambdapic_amd_$lib.so; fi
  n=0
  for set in "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
             "SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" \
             "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_VALU GRBM_GUI_ACTIVE"; do
    n=$((n+1))
    timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/${lib}_$n -- python3 $ROOT/bench.py --no-cpu-baseline --no-extra --steps 4 --warmup 2 > $OUT/${lib}_$n.log 2>&1
    echo "== $lib pass $n"; python3 $ROOT/tools/pmc_summary.py $OUT/${lib}_$n k_push_deposit_tiled_2d | grep -A12 "false, f" | head -12
  done
done
