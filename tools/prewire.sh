#!/bin/bash
# what the slab path costs before any wire: ONE 512x512 slab of C4's size (16 ppc) and ONE
# 64x256x256 slab of C5's, as rank 0 of a mirrored 2-slab ring (every kernel of the N > 1 path, device copies for the
# wire) against the same slab on the N = 1 path.  One gpurun call.
mkdir -p gpurun_out
out=gpurun_out/r03_prewire_slabs.txt
{
echo "== 2-D 512x512, 16 ppc (one of C4's eight slabs, uniform): N = 1 path"
python bench.py --nx 512 --ny 512 --ppc 16 --no-extra --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print(json.dumps({k:d[k] for k in ('ms_per_step','value')}), d['roofline']['kernel_ms'])"
echo "== the same slab as rank 0 of a mirrored 2-slab ring (in line / overlapped)"
python tools/bench_mirror.py --nx 512 --ny 512 --ppc 16 --steps 200 --warmup 20 2>/dev/null
python tools/bench_mirror.py --nx 512 --ny 512 --ppc 16 --steps 200 --warmup 20 --overlap 2>/dev/null
echo "== 3-D 64x256x256, 8 ppc (one of C5's eight slabs, uniform): N = 1 path"
python tools/bench_mirror3d.py --single --steps 40 --warmup 12 2>/dev/null
echo "== the same slab as rank 0 of a mirrored 2-slab ring (in line / overlapped)"
python tools/bench_mirror3d.py --steps 40 --warmup 12 2>/dev/null
python tools/bench_mirror3d.py --overlap --steps 40 --warmup 12 2>/dev/null
} > $out 2>&1
cat $out
