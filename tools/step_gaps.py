#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace of bench.py: the idle time between the end of one Lloyd iteration's last kernel
(k_mstep_flags) and the start of the next iteration's E-step (k_assign_mfma), and the step period (development aid).
Usage: tools/step_gaps.py <rocprof output dir>"""
import csv
import glob
import sys

import numpy as np

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
gaps, periods, last_flags_end, last_assign_start, between = [], [], None, None, []
for st, en, name in rows:
    if "k_assign_mfma" in name:
        if last_flags_end is not None:
            gaps.append((st - last_flags_end) / 1e3)
            between.append(cur_between)
        if last_assign_start is not None:
            periods.append((st - last_assign_start) / 1e6)
        last_assign_start, last_flags_end = st, None
    elif "k_mstep_flags" in name:
        last_flags_end, cur_between = en, []
    elif last_flags_end is not None:
        cur_between.append(name.split("(")[0][:40])
gaps, periods = np.array(gaps), np.array(periods)
# the timed region: the longest run of equal-length steps
print(f"iterations seen: {len(periods) + 1}")
print(f"k_mstep_flags end -> next k_assign_mfma start: median {np.median(gaps):.1f} us, p90 {np.percentile(gaps, 90):.1f} us, "
      f"max {gaps.max():.1f} us (n={len(gaps)})")
print(f"step period (E-step start to E-step start): median {np.median(periods):.4f} ms, min {periods.min():.4f} ms")
names = sorted({n for b in between for n in b})
print("kernels between the two:", names if names else "none")
