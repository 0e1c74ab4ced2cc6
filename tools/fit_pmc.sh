#!/bin/bash
# SQ counters of one reference-sized batched fit (eth), own pass (no tracing).
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
rm -rf $OUT/fit_pmc
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/fit_pmc -- python3 tools/profile_fit.py eth > $OUT/fit_pmc.log 2>&1 || { tail -20 $OUT/fit_pmc.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/fit_pmc/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    if "kpp_dist" in k or "assign" in k:
        print(k, {n: round(sum(v) / len(v)) for n, v in c.items()}, "launches", len(next(iter(c.values()))))
PY
