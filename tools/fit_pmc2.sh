#!/bin/bash
# Counters of one reference-sized batched fit, own pass (counters only, no tracing).
# Usage: tools/fit_pmc2.sh <tag> <config> COUNTER [COUNTER ...]     (at most 8 SQ_ counters per pass)
set -u
TAG=$1; CFG=$2; shift 2
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
rm -rf $OUT/fit_pmc_$TAG
timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/fit_pmc_$TAG -- python3 tools/profile_fit.py $CFG > $OUT/fit_pmc_$TAG.log 2>&1 || { tail -20 $OUT/fit_pmc_$TAG.log; exit 1; }
python3 - $TAG <<'PY' > gpurun_out/fit_pmc_$TAG.txt
import csv, glob, collections, sys
tag = sys.argv[1]
f = glob.glob(f"gpurun_out/fit_pmc_{tag}/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:30]
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in sorted(agg.items(), key=lambda kv: -sum(sum(v) for v in kv[1].values())):
    if not k.startswith("k_"):
        continue
    n = max(len(v) for v in c.values())
    print(f"{k:32s} launches {n:4d}  " + "  ".join(f"{name} {sum(v)/len(v):.4g}" for name, v in sorted(c.items())))
PY
cat gpurun_out/fit_pmc_$TAG.txt
