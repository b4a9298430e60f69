#!/bin/bash
# Per-WORKLOAD rocprofv3 evidence (run through gpurun from the repo root):   profiles/collect_workloads.sh r03 [workload ...]
# Every (problem, scheme, pattern, kernel) is profiled in a process of its own, so a kernel-name row of the summary is ONE
# workload (the bench command runs the same instantiation on the manual and the optimized pattern: its per-name rows mix them).
# Three passes per workload, never combined: --kernel-trace --stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE.
# -> gpurun_out/wl_<tag>/{<workload>.stats.csv, workloads.json}; workloads.json is copied to profiles/<tag>_workloads.json
set -e
cd "$(dirname "$0")/.."
TAG=${1:-r03}
shift || true
WL=${@:-cfg2 cfg3 cfg4 cfg4:optimized cfg5 cfg5:optimized cfg5p cfg5p:optimized q12_mid cfg2:hess cfg4:hess cfg5p:hess cfg5:hess q12_mid:hess cfg4:optimized:hess cfg5p:optimized:hess cfg5:optimized:hess q12_mid:optimized:hess cfg2:csr cfg4:csr cfg5:csr cfg5:optimized:csr}
OUT=gpurun_out/wl_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for w in $WL; do
    f=$(echo $w | tr ':' '_')
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$f.stats -- python3 bench/one_workload.py $w 300 > $OUT/$f.json 2> $OUT/$f.stats.log
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/$f.fetch -- python3 bench/one_workload.py $w 60 > /dev/null 2> $OUT/$f.fetch.log
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/$f.write -- python3 bench/one_workload.py $w 60 > /dev/null 2> $OUT/$f.write.log
    echo "profiled $w"
done
python3 - "$OUT" $WL <<'PY'
import csv, glob, json, sys
out, wls = sys.argv[1], sys.argv[2:]
rows = []
for w in wls:
    f = w.replace(":", "_")
    meta = json.loads([l for l in open(f"{out}/{f}.json") if l.startswith("{")][-1])
    want = "hess" if meta["kernel"] == "hess" else "cons_jac_kernel"
    stat = None
    for p in glob.glob(f"{out}/{f}.stats/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            if "ctd::" in r["Name"] and want in r["Name"] and "finish" not in r["Name"]:
                if stat is None or int(r["Calls"]) > int(stat["Calls"]):
                    stat = r
    pmc = {}
    for key, d in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
        vals = []
        for p in glob.glob(f"{out}/{f}.{d}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(p)):
                if r["Counter_Name"] == key and stat and r["Kernel_Name"] == stat["Name"]:
                    vals.append(float(r["Counter_Value"]))
        pmc[key] = sum(vals) / len(vals) if vals else None
    avg_ns = float(stat["AverageNs"]) if stat else None
    alg = meta["algorithmic_bytes_per_launch"]
    # gfx950: FETCH_SIZE tallies 128-byte read requests at 64 bytes -> x 2; WRITE_SIZE exact; both in KiB (MI355X_MICROARCH.md)
    rd = pmc["FETCH_SIZE"] * 1024 * 2 if pmc["FETCH_SIZE"] is not None else None
    wr = pmc["WRITE_SIZE"] * 1024 if pmc["WRITE_SIZE"] is not None else None
    traffic = rd + wr if rd is not None and wr is not None else None
    rows.append({**meta, "kernel_name": stat["Name"] if stat else None, "calls": int(stat["Calls"]) if stat else 0,
                 "rocprof_avg_ns": avg_ns, "rocprof_min_ns": float(stat["MinNs"]) if stat else None, "rocprof_max_ns": float(stat["MaxNs"]) if stat else None,
                 "achieved_GBs": alg / avg_ns if avg_ns else None, "frac_of_8TBs": alg / avg_ns / 8000.0 if avg_ns else None,
                 "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": traffic,
                 "traffic_over_algorithmic": traffic / alg if traffic else None})
json.dump({"what": "one process per workload; rocprofv3 --kernel-trace --stats (300 launches) and separate --pmc FETCH_SIZE / WRITE_SIZE passes (60 launches); "
                   "FETCH_SIZE x 2 on gfx950, WRITE_SIZE exact, KiB; frac = algorithmic bytes / rocprof average duration / 8 TB/s",
           "workloads": rows}, open(out + "/workloads.json", "w"), indent=1)
for r in rows:
    print(f"{r['workload']:18s} {r['rocprof_avg_ns'] or 0:10.0f} ns  frac {r['frac_of_8TBs'] or 0:.3f}  traffic/alg {r['traffic_over_algorithmic'] or 0:.2f}  {r['kernel_name']}")
PY
cp $OUT/workloads.json profiles/${TAG}_workloads.json 2>/dev/null || true
