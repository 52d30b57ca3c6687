#!/bin/bash
# usage: tools/pmc3d_lds.sh <tag> <variant|product> [bench3d args]: LDS counters of K1-3D for one build / order
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; var=$2; shift 2
unset LPA_LIB_PATH
[ "$var" != product ] && export LPA_LIB_PATH=$ROOT/lambdapic_amd/csrc/build/liblambdapic_amd_$var.so
OUT=$ROOT/gpurun_out/pmc3dl_$tag; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES \
   --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/tools/bench3d.py --steps 10 --warmup 2 "$@" > $OUT.log 2>&1
echo "== $tag (exit $?)"
python3 $ROOT/tools/pmc_summary.py $OUT k_push_deposit_tiled_3d
