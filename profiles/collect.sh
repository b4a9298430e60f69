#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   profiles/collect.sh r01
# 1. --kernel-trace --stats over the exact bench command            -> gpurun_out/prof_<tag>/bench_stats
# 2. separate --pmc passes for FETCH_SIZE and WRITE_SIZE (never combined with other trace domains)
# 3. the PCIe-inclusive (host-pointer) rate of the same workload, which is never `value`
# then summarises into gpurun_out/prof_<tag>/{summary.json, pmc_traffic.json, kernel_stats.csv} (copied to profiles/ and
# committed).
set -e
cd "$(dirname "$0")/.."
TAG=${1:-r03}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 500 --warmup 50 --no-cpu-baseline --no-extras"      # (--no-extras: the trace then holds ONLY the timed kernel -- the extras launch the same template on the CSR order and other sizes)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_stats -- $BENCH > $OUT/bench_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, json, sys, collections, shutil
out = sys.argv[1]
res = {"command": "python3 bench.py --steps 500 --warmup 50 --no-cpu-baseline --no-extras", "kernels": {}}
for f in glob.glob(out + "/bench_stats/**/*kernel_stats.csv", recursive=True):
    shutil.copy(f, out + "/kernel_stats.csv")
    for r in csv.DictReader(open(f)):
        if "ctd::" in r["Name"]:
            res["kernels"][r["Name"]] = {k: r[k] for k in ("Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "StdDev", "Percentage")}
pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("pmc_fetch", "pmc_write"):
    for f in glob.glob(out + f"/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "ctd::cons_jac_kernel" in r["Kernel_Name"] or "ctd::hess_kernel" in r["Kernel_Name"] or "ctd::hess_step_kernel" in r["Kernel_Name"]:
                pmc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res["pmc_mean_per_launch"] = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in pmc.items()}
json.dump(res, open(out + "/summary.json", "w"), indent=1)
# HBM traffic per launch, corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE (KiB) x 2
# (128-B read requests are tallied at 64 B), WRITE_SIZE (KiB) exact
traffic = {"bench_kernel": None, "other_kernels": {}, "hessian_kernels": {},
           "correction": "FETCH_SIZE x 2 (gfx950 tallies 128-B read requests at 64 B), WRITE_SIZE exact; separate --pmc passes; "
                         "profiles/collect.sh"}
for name, c in res["pmc_mean_per_launch"].items():
    rd, wr = c.get("FETCH_SIZE", 0.0) * 1024 * 2, c.get("WRITE_SIZE", 0.0) * 1024
    e = {"kernel": name, "FETCH_SIZE_KiB_raw": c.get("FETCH_SIZE"), "WRITE_SIZE_KiB_raw": c.get("WRITE_SIZE"),
         "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
         "rocprof_avg_kernel_ns": float(res["kernels"].get(name, {}).get("AverageNs", "nan"))}
    if "hess_kernel" in name or "hess_step_kernel" in name: traffic["hessian_kernels"][name] = e
    elif "cons_jac_kernel<ctd::GoddardOCP, 2, 2" in name: traffic["bench_kernel"] = e
    else: traffic["other_kernels"][name] = e
json.dump(traffic, open(out + "/pmc_traffic.json", "w"), indent=1)
print(json.dumps(res["kernels"], indent=1)[:3000])
PY
python3 - "$OUT" <<'PY'
# PCIe-inclusive rate: the host-pointer entry point ctd_cons_jac (H2D of x, kernel, D2H of c and the values) on the bench workload
import json, sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import ctdirect_jl_amd as ct
from helpers import bench_inputs, describe
d = ct.DOCP("goddard", 10000, "gauss_legendre_2", device=0)
x = bench_inputs(describe(d, "goddard", "gauss_legendre_2"), perturb=1e-3)
c = np.empty(d.dim_NLP_constraints); v = np.empty(d.nnzj)
for _ in range(20): d.cons_jac(x, c, v)
t0 = time.perf_counter(); K = 200
for _ in range(K): d.cons_jac(x, c, v)
el = (time.perf_counter() - t0) / K
res = {"workload": "goddard/gauss_legendre_2 N=10000, host pointers (pageable numpy arrays): H2D x + kernel + D2H c, vals",
       "ms_per_eval": el * 1e3, "evals_per_s": 1.0 / el, "bytes_over_pcie": 8 * (d.dim_NLP_variables + d.dim_NLP_constraints + d.nnzj)}
# the same call on page-locked arrays (ctd_host_alloc): direct DMA instead of the runtime's bounce buffers
xp, cp, vp = ct.pinned_empty(x.size), ct.pinned_empty(c.size), ct.pinned_empty(v.size)
xp[:] = x
for _ in range(20): d.cons_jac(xp, cp, vp)
t0 = time.perf_counter()
for _ in range(K): d.cons_jac(xp, cp, vp)
elp = (time.perf_counter() - t0) / K
assert np.array_equal(cp, c) and np.array_equal(vp, v)
res["pinned"] = {"ms_per_eval": elp * 1e3, "evals_per_s": 1.0 / elp, "GBs_over_pcie": res["bytes_over_pcie"] / elp / 1e9}
json.dump(res, open(sys.argv[1] + "/host_pointer_rate.json", "w"), indent=1)
print(json.dumps(res))
PY
