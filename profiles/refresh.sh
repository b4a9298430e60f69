#!/bin/bash
# Re-collects every piece of measured evidence of a round on the GPU box (run through gpurun from the repo root):
#   profiles/refresh.sh r02     -> gpurun_out/evidence_<tag>/  (copy what is to be judged into profiles/)
# bench lines (driver-style 20 steps, default, 2000 steps), rocprofv3 stats + HBM counters + SQ counters of the bench command,
# in-kernel phase stamps of both kernel families, tile sweeps of the Hessian kernel, whole-iteration timings.
set -e
cd "$(dirname "$0")/.."
TAG=${1:-r02}
E=gpurun_out/evidence_$TAG
mkdir -p $E
python3 bench.py --steps 20 --warmup 5 > $E/bench_line_steps20.json 2> $E/bench_steps20.err
python3 bench.py > $E/bench_line_default.json 2> $E/bench_default.err
python3 bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-extras > $E/bench_line_steps2000.json 2> $E/bench_steps2000.err
echo "bench lines done"
bash profiles/collect.sh $TAG > $E/collect.log 2>&1
echo "rocprof + HBM counters done"
bash profiles/collect_sq.sh $TAG > $E/collect_sq.log 2>&1
echo "SQ counters done"
python3 bench/stamps.py > $E/phase_stamps.log 2>&1
python3 bench/hess_stamps.py > $E/hessian_phase_stamps.log 2>&1
python3 bench/hess_tile_sweep.py cfg2 cfg4 cfg5p cfg5 > $E/hessian_tiles.log 2>&1
python3 bench/iteration.py > $E/iteration.log 2>&1
echo "stamps, sweeps, iteration done"
