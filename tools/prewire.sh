#!/bin/bash
# what the slab path costs before any wire: ONE slab as rank 0 of a mirrored 2-slab ring (every kernel of the N > 1 path;
# the wire = python-side copies between lpa_step sub-ranges / one copy kernel per round inside lpa_step / a one-rank RCCL
# communicator sending to itself) against the same slab on the N = 1 path; --b-messages: the four-round protocol (B guard
# planes exchanged) against the default two rounds.  Slabs: 512x512 at 16 ppc (one of C4's eight),
# 128x1024 at 64 ppc (C2 / 8: north_star's ">= 6x at 8 GPUs" case), 1024x1024 at 64 ppc (the weak-scaled headline's slab),
# 64x256x256 at 8 ppc (one of C5's eight).  One gpurun call.
mkdir -p gpurun_out
out=gpurun_out/r04_prewire_slabs.txt
one() { python bench.py --nx $1 --ny $2 --ppc $3 --no-extra --no-cpu-baseline --steps 200 --warmup 40 $4 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print(json.dumps({k:d[k] for k in ('ms_per_step','value')}), 'K1', d['roofline']['kernel_ms'])"; }
mir() { python tools/bench_mirror.py --nx $1 --ny $2 --ppc $3 --steps 200 --warmup 40 "${@:4}" 2>&1 | tail -1; }
{
for cfg in "512 512 16" "128 1024 64" "1024 1024 64"; do
  set -- $cfg
  echo "== 2-D $1x$2, $3 ppc: N = 1 path (step / run_steps)"
  one $1 $2 $3
  one $1 $2 $3 --run-steps
  echo "== the same slab as rank 0 of a mirrored 2-slab ring"
  mir $1 $2 $3 --transport python
  mir $1 $2 $3 --transport loopback
  mir $1 $2 $3 --transport loopback --run-steps
  mir $1 $2 $3 --transport loopback --overlap
  mir $1 $2 $3 --transport rccl
  mir $1 $2 $3 --transport rccl --run-steps
  mir $1 $2 $3 --transport loopback --run-steps --overlap
  mir $1 $2 $3 --transport rccl --run-steps --overlap
  mir $1 $2 $3 --transport loopback --run-steps --b-messages
  mir $1 $2 $3 --transport rccl --run-steps --b-messages
done
echo "== 3-D 64x256x256, 8 ppc (one of C5's eight slabs, uniform): N = 1 path"
python tools/bench_mirror3d.py --single --steps 40 --warmup 12 2>&1 | tail -1
echo "== the same slab as rank 0 of a mirrored 2-slab ring"
for t in python loopback rccl; do python tools/bench_mirror3d.py --steps 40 --warmup 12 --transport $t 2>&1 | tail -1; done
python tools/bench_mirror3d.py --steps 40 --warmup 12 --transport loopback --run-steps 2>&1 | tail -1
python tools/bench_mirror3d.py --steps 40 --warmup 12 --transport python --overlap 2>&1 | tail -1
python tools/bench_mirror3d.py --steps 40 --warmup 12 --transport loopback --overlap 2>&1 | tail -1
python tools/bench_mirror3d.py --steps 40 --warmup 12 --transport rccl --overlap 2>&1 | tail -1
python tools/bench_mirror3d.py --steps 40 --warmup 12 --transport loopback --run-steps --overlap 2>&1 | tail -1
python tools/bench_mirror3d.py --steps 40 --warmup 12 --transport rccl --run-steps --overlap 2>&1 | tail -1
python tools/bench_mirror3d.py --steps 40 --warmup 12 --transport rccl --run-steps 2>&1 | tail -1
python tools/bench_mirror3d.py --steps 40 --warmup 12 --transport loopback --b-messages 2>&1 | tail -1
python tools/bench_mirror3d.py --steps 40 --warmup 12 --transport rccl --b-messages 2>&1 | tail -1
} > $out 2>&1
cat $out
