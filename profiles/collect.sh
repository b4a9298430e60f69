#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   profiles/collect.sh r01
# 1. --kernel-trace --stats over the exact bench command            -> gpurun_out/prof_<tag>/bench_stats
# 2. separate --pmc passes for FETCH_SIZE and WRITE_SIZE (never combined with other trace domains)
# then summarises into gpurun_out/prof_<tag>/summary.json (copied to profiles/ by hand and committed).
set -e
cd "$(dirname "$0")/.."
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 500 --warmup 50 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_stats -- $BENCH > $OUT/bench_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
res = {"command": "python3 bench.py --steps 500 --warmup 50 --no-cpu-baseline", "kernels": {}}
for f in glob.glob(out + "/bench_stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "ctd::" in r["Name"]:
            res["kernels"][r["Name"]] = {k: r[k] for k in ("Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "StdDev", "Percentage")}
pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("pmc_fetch", "pmc_write"):
    for f in glob.glob(out + f"/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "ctd::cons_jac_kernel" in r["Kernel_Name"]:
                pmc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res["pmc_mean_per_launch"] = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in pmc.items()}
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
