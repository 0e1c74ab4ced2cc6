#!/bin/bash
# SQ counters of one reference-sized batched fit (eth), own pass (counters only, no tracing).
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
rm -rf $OUT/fit_pmc
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/fit_pmc -- python3 tools/profile_fit.py eth > $OUT/fit_pmc.log 2>&1 || { tail -20 $OUT/fit_pmc.log; exit 1; }
python3 - <<'PY' > gpurun_out/fit_eth_pmc.txt
import csv, glob, collections
f = glob.glob("gpurun_out/fit_pmc/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:44]
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("SQ counters per launch, one batched eth fit (N=73568, D=96, K=200, 20 restarts), averages over live launches;")
print("mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES/1024 SIMDs / (GRBM_GUI_ACTIVE/8 XCDs)")
for k, c in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("GRBM_GUI_ACTIVE", [0]))):
    if not k.startswith("k_"):
        continue
    gui = c.get("GRBM_GUI_ACTIVE", [0]); live = [i for i, g in enumerate(gui) if g > 0.02 * max(gui)] if max(gui) > 0 else []
    if not live:
        continue
    avg = {n: sum(v[i] for i in live) / len(live) for n, v in c.items()}
    busy = (avg.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024) / (avg["GRBM_GUI_ACTIVE"] / 8) if avg.get("GRBM_GUI_ACTIVE") else 0
    print(f"{k:46s} launches {len(live):4d}  mfma_busy {busy:5.3f}  gui_active/XCD {avg['GRBM_GUI_ACTIVE']/8:10.0f}  "
          f"lds_bank_conflict {avg.get('SQ_LDS_BANK_CONFLICT',0):10.0f}  wait_inst_any {avg.get('SQ_WAIT_INST_ANY',0):12.0f}  wave_cycles {avg.get('SQ_WAVE_CYCLES',0):12.0f}")
PY
cat gpurun_out/fit_eth_pmc.txt
