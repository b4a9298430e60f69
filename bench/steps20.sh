#!/bin/bash
# the driver's command form (20 timed steps): evals/s with and without a runtime setting, alternating:
#   bash bench/steps20.sh HSA_ENABLE_INTERRUPT=0        bash bench/steps20.sh ROC_ACTIVE_WAIT_TIMEOUT=1000
SETTING=${1:-HSA_ENABLE_INTERRUPT=0}
for rep in 1 2 3 4; do
  for mode in default "$SETTING"; do
    if [ "$mode" = default ]; then v=$(python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null); else v=$(env $mode python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null); fi
    echo "$mode $(echo "$v" | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), round(d["ms_per_step"]*1e3,3), round(d["roofline"]["kernel_ms"]*1e3,3))')"
  done
done
