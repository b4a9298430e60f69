#!/bin/bash
# Wave-level issue / stall counters of the bench command (run through gpurun from the repo root, after collect.sh):
#   profiles/collect_sq.sh r01
# Two --pmc passes of 8 SQ counters each (the slot budget of gfx950, /opt/skills/guides/MI355X_MICROARCH.md "rocprofv3
# PMC slots"), kernel trace only.  Summary -> gpurun_out/prof_<tag>/sq_counters.json (copied to profiles/ and committed).
# WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES: where a wave's time goes (parked on s_waitcnt / barrier, stalled at
# issue, issuing).
set -e
cd "$(dirname "$0")/.."
TAG=${1:-r02}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT \
  --kernel-trace --output-format csv -d $OUT/pmc_sq1 -- $BENCH > $OUT/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS \
  --kernel-trace --output-format csv -d $OUT/pmc_sq2 -- $BENCH > $OUT/pmc_sq2.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("pmc_sq1", "pmc_sq2"):
    for f in glob.glob(out + f"/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "ctd::cons_jac_kernel" in r["Kernel_Name"] or "ctd::hess_kernel" in r["Kernel_Name"] or "ctd::hess_step_kernel" in r["Kernel_Name"]:
                pmc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {"command": "python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline",
       "note": "mean per launch over all launches of the kernel; SQ_*_CYCLES / WAIT / ACTIVE counters are in quad-cycles summed "
               "over waves; fractions are of SQ_WAVE_CYCLES", "kernels": {}}
for k, d in pmc.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    wc = m.get("SQ_WAVE_CYCLES", 0.0) or float("nan")
    m["frac_wait_any(parked: s_waitcnt / barrier)"] = m.get("SQ_WAIT_ANY", 0.0) / wc
    m["frac_wait_inst_any(issue stall)"] = m.get("SQ_WAIT_INST_ANY", 0.0) / wc
    m["frac_active_inst_any(issuing)"] = m.get("SQ_ACTIVE_INST_ANY", 0.0) / wc
    m["frac_active_inst_valu"] = m.get("SQ_ACTIVE_INST_VALU", 0.0) / wc
    if m.get("SQ_LDS_IDX_ACTIVE"):
        m["lds_bank_conflict_frac_of_lds_cycles"] = m.get("SQ_LDS_BANK_CONFLICT", 0.0) / m["SQ_LDS_IDX_ACTIVE"]
    if m.get("SQ_WAVES"):
        m["valu_insts_per_wave"] = m.get("SQ_INSTS_VALU", 0.0) / m["SQ_WAVES"]
    res["kernels"][k] = m
json.dump(res, open(out + "/sq_counters.json", "w"), indent=1)
for k, m in res["kernels"].items():
    print(k[:80], {c: round(v, 3) for c, v in m.items() if c.startswith(("frac", "lds_", "valu_"))})
PY
