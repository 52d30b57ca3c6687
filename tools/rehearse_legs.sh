#!/bin/bash
# rehearsal of bench.py --gpus N on ONE GPU: N ranks share cuda:0 and talk through gloo (faces staged through the host).
# The numbers mean nothing; what counts is that every leg runs and its bookkeeping assertions hold.
# usage: rehearse_legs.sh N [legs] [leg-steps] [extra bench.py flags ...]   (e.g. "--backend nccl": the RCCL pre-flights fail
# on a shared GPU -- "Duplicate GPU detected" -- and the run must fall back to gloo; "--stall-rank 1 --leg-timeout 60": a rank
# that stops inside the C4 leg must leave the headline line, a partial final line and a non-zero exit)
N=${1:-2}; LEGS=${2:-c2s,c4,c5}; STEPS=${3:-12}; shift 3 2>/dev/null
TAG=r04_rehearse_n$N$(echo "$*" | tr -c 'a-z0-9\n' '_' | cut -c1-40)
mkdir -p gpurun_out
BACKEND="--backend gloo"; case "$*" in *--backend*) BACKEND="";; esac
python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus $N --steps 4 --warmup 2 $BACKEND --share-gpu --leg-steps $STEPS --legs $LEGS "$@" \
  > gpurun_out/$TAG.json.log 2> gpurun_out/$TAG.err
rc=$?
echo "exit code $rc" | tee -a gpurun_out/$TAG.err
grep "\[bench\]" gpurun_out/$TAG.err | tail -8
python - <<PY
import json
lines = [l for l in open("gpurun_out/$TAG.json.log").read().strip().splitlines() if l.startswith("{")]
print(len(lines), "JSON line(s)")
for k, l in enumerate(lines):
    d = json.loads(l)
    print(f"line {k}: headline n_gpus", d["n_gpus"], "value %.4g" % d["value"], "| comm:", d["config"]["comm"], "| world", d["config"].get("world"),
          "| rccl", d["config"].get("rccl_version"))
    for e in d.get("extra", []):
        print("   leg:", e.get("workload", "")[:60], "| value", e.get("value"), "| ms", e.get("ms_per_step"), "| err", e.get("error"),
              "| charge", e.get("charge_rel_err"), "| alive_per_rank", e.get("alive_per_rank"), "| ledger", e.get("ledger"))
PY
exit 0
