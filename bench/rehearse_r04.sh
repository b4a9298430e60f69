#!/bin/bash
# Round-4 rehearsal of the multi-GPU bench on the ONE GPU of a gpurun box (the ranks share the device, gloo carries the
# collectives): (1) the default self-launched line with its `strong` block, (2) the same through torch.distributed.run as the
# driver starts it, (3) --x-mode halo, (4) the fail-fast hook: rank 1 dies before the first collective.
cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
set -o pipefail
t0=$SECONDS
timeout -k 10 420 python bench.py --gpus 2 --steps 200 --warmup 20 > $O/r04_2rank_selflaunch.json 2> $O/r04_2rank_selflaunch.err; echo "selflaunch rc=$? $((SECONDS - t0)) s"
t0=$SECONDS
timeout -k 10 420 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 2 --steps 200 --warmup 20 > $O/r04_2rank_torchrun.json 2> $O/r04_2rank_torchrun.err; echo "torchrun rc=$? $((SECONDS - t0)) s"
timeout -k 10 300 python bench.py --gpus 2 --steps 200 --warmup 20 --x-mode halo --no-strong > $O/r04_2rank_halo.json 2> $O/r04_2rank_halo.err; echo "halo rc=$?"
t0=$SECONDS
CTD_BENCH_TEST_EXIT_RANK=1 timeout -k 10 300 python bench.py --gpus 2 --steps 200 --warmup 20 > $O/r04_2rank_kill.json 2> $O/r04_2rank_kill.err; echo "kill-hook rc=$? (expected 17) $((SECONDS - t0)) s total (includes the ranks' start-up)"
grep "launcher:" $O/r04_2rank_kill.err
python - <<'PY'
import json
for f in ("r04_2rank_selflaunch", "r04_2rank_torchrun", "r04_2rank_halo"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/{f}.json") if l.startswith("{")][-1])
    except Exception as e:
        print(f, "NO LINE", e); continue
    print(f, "| value", round(d["value"]), "| ms/step", round(d["ms_per_step"], 4), "| x_mode", d["config"].get("x_mode"),
          "| check", d.get("sharded_iterate_check", {}).get("bit_identical_to_whole_iterate_on_every_rank"),
          "|", {k: round(d[k]["ms_per_step"], 4) for k in ("halo_allgather", "peer_in_place", "stitched_c", "ordered_step", "broadcast_x", "no_exchange") if k in d and "ms_per_step" in d[k]})
    si = d.get("sharded_iteration", {})
    print("    sharded_iteration |", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in si.items() if k != "what"})
    for k, b in d.get("strong", {}).items():
        if "value" not in b:
            print("   ", k, b); continue
        print("   ", k, "| evals/s", round(b["value"]), "| ms/step", round(b["ms_per_step"], 4), "| x_mode", b["x_mode"],
              "| per-rank us/frac", [(round(r["kernel_ms"] * 1e3, 2), round(r["frac"], 3)) for r in b["per_rank"]],
              "| stitched", round(b.get("stitched_c", {}).get("ms_per_step", -1), 4), "| ordered", round(b.get("ordered_step", {}).get("ms_per_step", -1), 4),
              "| iteration", round(b.get("sharded_iteration", {}).get("ms_per_iteration", -1), 4),
              "| check", b.get("sharded_iterate_check", {}).get("bit_identical_to_whole_iterate_on_every_rank"))
PY
