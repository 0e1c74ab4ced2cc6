#!/usr/bin/env python3
"""Print our kernels' rows of a rocprofv3 --stats kernel_stats.csv (development aid)."""
import csv
import glob
import sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "k_" in r["Name"]:
        print(r["Name"].split("(")[0][:44].ljust(45), r["Calls"].rjust(6), f"{float(r['TotalDurationNs'])/1e6:9.3f} ms",
              f"{float(r['AverageNs'])/1e3:9.2f} us avg", f"min {float(r['MinNs'])/1e3:8.2f} max {float(r['MaxNs'])/1e3:8.2f}")
