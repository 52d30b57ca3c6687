#!/bin/bash
# usage: tools/pmc3d_one.sh <tag> [variant]: LDS counters of K1-3D (tools/bench3d.py, default sort interval)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc3d_$1
[ -n "$2" ] && export LPA_LIB_PATH=$ROOT/lambdapic_amd/csrc/build/liblambdapic_amd_$2.so
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL \
   --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/tools/bench3d.py --steps 10 --warmup 2 > $OUT.log 2>&1
echo "== $1 (exit $?)"
python3 $ROOT/tools/pmc_summary.py $OUT k_push_deposit_tiled_3d
