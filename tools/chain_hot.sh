#!/bin/bash
# a hot plasma on a slab chain (2 ranks sharing ONE GPU through gloo: absolute numbers mean nothing, the ratio does): the
# chain's common sort clock on the fixed interval against the interval the ranks agree on from their overflow lists
for u in 0.2 0.5; do for f in "--fixed-sort" ""; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 \
    bench.py --gpus 2 --steps 60 --warmup 8 --backend gloo --share-gpu --legs none --nx 512 --uth $u $f 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('uth=$u $f ms/step %.3f value %.3e' % (d['ms_per_step'], d['value']))
"
done; done
