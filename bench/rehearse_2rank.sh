#!/bin/bash
# Two ranks on the ONE GPU of a gpurun box (gloo carries the collectives; RCCL refuses two ranks on one device): the
# torch.distributed.run launch of bench.py for every --config, as the driver launches it on an 8-GPU node.
cd "$(dirname "$0")/.."
out=${1:-gpurun_out/r02_rehearsal_2rank.jsonl}
: > "$out"
for cfg in cfg2_weak cfg2_strong cfg4 cfg5; do
  CTD_BENCH_DEVICE=0 CTD_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
      --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 200 --warmup 20 --config $cfg 2>gpurun_out/rehearsal_$cfg.err | grep '^{' >> "$out" \
      || { echo "rehearsal $cfg failed"; tail -5 gpurun_out/rehearsal_$cfg.err; }
done
python - "$out" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    d = json.loads(line)
    print(d["config"]["workload"][:60], "| value", round(d["value"]), d["unit"], "| ms/step", round(d["ms_per_step"], 4),
          "| per-rank", [(r["rank"], round(r["kernel_ms"] * 1e3, 2), round(r["frac"], 3)) for r in d["roofline"]["per_rank"]],
          "| stitched", round(d["stitched_c"].get("ms_per_step", -1), 4) if "stitched_c" in d else None,
          "| bcast", round(d["broadcast_x"].get("ms_per_step", -1), 4) if "broadcast_x" in d else None,
          "| noex", round(d["no_exchange"].get("ms_per_step", -1), 4) if "no_exchange" in d else None)
PY
