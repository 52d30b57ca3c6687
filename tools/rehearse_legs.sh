#!/bin/bash
# rehearsal of bench.py --gpus N on ONE GPU: N ranks share cuda:0 and talk through gloo (faces staged through the host).
# The numbers mean nothing; what counts is that every leg runs and its bookkeeping assertions hold.  usage: rehearse_legs.sh N [legs]
N=${1:-2}; LEGS=${2:-c2s,c4,c5}
mkdir -p gpurun_out
python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus $N --steps 4 --warmup 2 --backend gloo --share-gpu --leg-steps ${3:-12} --legs $LEGS \
  > gpurun_out/r03_rehearse_n$N.json.log 2> gpurun_out/r03_rehearse_n$N.err
rc=$?
tail -c 1500 gpurun_out/r03_rehearse_n$N.err
python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r03_rehearse_n$N.json.log").read().strip().splitlines()[-1])
    print("headline", d["n_gpus"], d["value"], d["config"]["comm"])
    for e in d.get("extra", []):
        print(" leg:", e.get("workload", "")[:60], "| value", e.get("value"), "| ms", e.get("ms_per_step"), "| err", e.get("error"),
              "| charge", e.get("charge_rel_err"), "| alive_per_rank", e.get("alive_per_rank"), "| ledger", e.get("ledger"))
except Exception as ex:
    print("no line:", ex)
PY
exit $rc
