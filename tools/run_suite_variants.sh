#!/bin/bash
# the whole -m gpu suite with lib() pointed at the VARIANTS build (LPA_LIB_PATH): the variant library must be a drop-in
set -o pipefail
mkdir -p gpurun_out
LPA_LIB_PATH=$PWD/lambdapic_amd/csrc/build/liblambdapic_amd_variants.so python -m pytest tests -x -q -m gpu > gpurun_out/${1:-r03}_gpu_tests_variants_lib.log 2>&1
rc=$?
tail -3 gpurun_out/${1:-r03}_gpu_tests_variants_lib.log
exit $rc
