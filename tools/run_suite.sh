#!/bin/bash
# full -m gpu suite + the default bench line in one gpurun call; logs under gpurun_out/ (tag = $1)
set -o pipefail
tag=${1:-r03}
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/${tag}_gpu_tests.log 2>&1
rc=$?
tail -3 gpurun_out/${tag}_gpu_tests.log
[ $rc -ne 0 ] && exit $rc
python bench.py > gpurun_out/${tag}_bench.json.log 2> gpurun_out/${tag}_bench.err
rc=$?
tail -c 2500 gpurun_out/${tag}_bench.json.log
exit $rc
