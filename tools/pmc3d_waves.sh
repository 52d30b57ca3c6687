#!/bin/bash
# usage: tools/pmc3d_waves.sh <tag> [variant]: where K1-3D's wave cycles go (two counter passes; tools/bench3d.py)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
[ -n "$2" ] && export LPA_LIB_PATH=$ROOT/lambdapic_amd/csrc/build/liblambdapic_amd_$2.so
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1)); OUT=$ROOT/gpurun_out/pmc3dw_$1_$i; mkdir -p $OUT
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/tools/bench3d.py --steps 10 --warmup 2 > $OUT.log 2>&1
  echo "== $1 pass $i (exit $?)"
  python3 $ROOT/tools/pmc_summary.py $OUT k_push_deposit_tiled_3d
done
