#!/bin/bash
# A/B: the leaver check of the push kernels (lpa_push_params.leavers) compiled out vs in, headline config + C2/8 slab
for i in 1 2; do
for lib in lambdapic_amd/csrc/build/liblambdapic_amd_noleavers.so lambdapic_amd/liblambdapic_amd.so; do
  echo "== $lib"
  LPA_LIB_PATH=$PWD/$lib python bench.py --no-extra --no-cpu-baseline --steps 40 --warmup 25 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('C2', d['ms_per_step'], 'K1', d['roofline']['kernel_ms'])"
  LPA_LIB_PATH=$PWD/$lib python bench.py --nx 128 --ny 1024 --no-extra --no-cpu-baseline --steps 200 --warmup 40 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('C2/8', d['ms_per_step'], 'K1', d['roofline']['kernel_ms'])"
done
done
