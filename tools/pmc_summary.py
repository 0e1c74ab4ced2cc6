#!/usr/bin/env python3
"""Condense the rocprofv3 outputs of tools/gpu_round.sh into what profiles/ keeps:

    tools/pmc_summary.py <tag> <round-name>      e.g.  tools/pmc_summary.py r01d r01

reads  gpurun_out/<tag>_prof/**/_kernel_stats.csv, gpurun_out/<tag>_pmc_{fetch,write,sq}/**/_counter_collection.csv
writes profiles/<round>_bench_kernel_stats.csv   (our kernels' rows of the --stats summary)
       profiles/<round>_pmc.json                 (per kernel: launches, average counter values, HBM bytes/launch)
       profiles/pmc_traffic.json                 (what bench.py reports as roofline.traffic for this workload)

HBM bytes per launch follow MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE and
WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts half the bytes of wide (16 B per lane)
streaming reads, so it is doubled for the kernels whose reads are 16 B per lane (all of ours
below); WRITE_SIZE is exact.  traffic = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024.
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OURS = ("k_assign", "k_accumulate", "k_best_of_k", "k_mstep", "k_inertia", "k_col_stats", "k_sub_rowvec", "k_argmax",
        "k_far", "k_apply_relocation", "k_stage_centers", "k_kpp", "k_ade_fde", "k_cluster_means", "k_row_sqnorm",
        "k_soft", "k_nms", "k_decode", "k_topm")


def short(name):
    n = name.split("(")[0].replace("void ", "").strip()
    return n


def counters(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        if any(o in k for o in OURS):
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def main():
    tag, rnd = sys.argv[1], sys.argv[2]
    out = os.path.join(ROOT, "gpurun_out")
    prof = os.path.join(ROOT, "profiles")
    stats = glob.glob(os.path.join(out, f"{tag}_prof", "**", "*_kernel_stats.csv"), recursive=True)[0]
    rows = list(csv.reader(open(stats)))
    with open(os.path.join(prof, f"{rnd}_bench_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(rows[0])
        for r in rows[1:]:
            if any(o in r[0] for o in OURS) or "copyBuffer" in r[0] or "fillBuffer" in r[0]:
                w.writerow(r)
    summary = {}
    for kind in ("fetch", "write", "sq"):
        files = glob.glob(os.path.join(out, f"{tag}_pmc_{kind}", "**", "*_counter_collection.csv"), recursive=True)
        if not files:
            continue
        for k, cs in counters(files[0]).items():
            e = summary.setdefault(k, {})
            for c, v in cs.items():
                # launches enqueued behind a paused / finished Lloyd iteration return at once (state != RUNNING):
                # they are not the kernel's work and are left out of the averages
                live = [x for x in v if x > 0.02 * max(v)] if max(v) > 0 else v
                e[c] = {"launches": len(live), "noop_launches_dropped": len(v) - len(live), "avg": sum(live) / len(live)}
    # The E-step of a single run is two launches per iteration since round 2 (whole rounds of 8-wave workgroups, then
    # the remaining points with wider ones: another instantiation, or the same one on a few workgroups): per
    # iteration = the mean of the big launches + the small ones' total spread over them.
    estep = {}
    for kind in ("fetch", "write"):
        files = glob.glob(os.path.join(out, f"{tag}_pmc_{kind}", "**", "*_counter_collection.csv"), recursive=True)
        if not files:
            continue
        per_d = collections.defaultdict(list)
        for k, cs in counters(files[0]).items():
            if k.startswith("k_assign_mfma<"):
                for c, v in cs.items():
                    per_d[(k.split(",")[0], c)] += v
        for (kd, c), v in per_d.items():
            big = [x for x in v if x > 0.5 * max(v)]
            small = [x for x in v if 1e-4 * max(v) < x <= 0.5 * max(v)]
            estep.setdefault(kd, {})[c] = sum(big) / len(big) + sum(small) / len(big)
    traffic = {}
    for k, e in summary.items():
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            e["hbm_bytes_per_launch"] = 2 * e["FETCH_SIZE"]["avg"] * 1024 + e["WRITE_SIZE"]["avg"] * 1024
            traffic[k.split("<")[0]] = e["hbm_bytes_per_launch"]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in e and e["SQ_VALU_MFMA_BUSY_CYCLES"]["avg"] > 0 and "GRBM_GUI_ACTIVE" in e:
            # MFMA busy cycles are summed over 1024 SIMDs, GRBM_GUI_ACTIVE over 8 XCDs
            e["mfma_busy_frac"] = (e["SQ_VALU_MFMA_BUSY_CYCLES"]["avg"] / 1024) / (e["GRBM_GUI_ACTIVE"]["avg"] / 8)
    bench = json.load(open(os.path.join(out, f"{tag}_pmc_fetch.json")))
    kd = "k_assign_mfma<%d" % bench["config"]["d"]
    if kd in estep and "FETCH_SIZE" in estep[kd] and "WRITE_SIZE" in estep[kd]:
        traffic["k_assign_mfma"] = 2 * estep[kd]["FETCH_SIZE"] * 1024 + estep[kd]["WRITE_SIZE"] * 1024
        summary[kd + ", ...> per iteration (all launches)"] = {"FETCH_SIZE": estep[kd]["FETCH_SIZE"],
                                                              "WRITE_SIZE": estep[kd]["WRITE_SIZE"],
                                                              "hbm_bytes_per_launch": traffic["k_assign_mfma"]}
    json.dump({"source": f"gpurun_out/{tag}_pmc_*", "bench_config": bench["config"], "kernels": summary},
              open(os.path.join(prof, f"{rnd}_pmc.json"), "w"), indent=1)
    json.dump({"round": rnd, "n": bench["config"]["n"], "d": bench["config"]["d"], "k": bench["config"]["k"],
               "n_gpus": 1, "eval": bench.get("best_of_k", {}).get("config"),
               "hbm_bytes_per_launch": traffic,
               "method": "2*FETCH_SIZE + WRITE_SIZE (KiB), separate --pmc passes, MI355X_MICROARCH.md HBM section"},
              open(os.path.join(prof, "pmc_traffic.json"), "w"), indent=1)
    for name in ("bench", "bench_prof"):
        src = os.path.join(out, f"{tag}_{name}.json")
        if os.path.exists(src):
            dst = {"bench": f"{rnd}_bench.json", "bench_prof": f"{rnd}_bench_under_rocprof.json"}[name]
            open(os.path.join(prof, dst), "w").write(open(src).read())
    for k, e in sorted(summary.items()):
        print(k, {c: (round(v["avg"], 1) if isinstance(v, dict) else round(v, 4)) for c, v in e.items()
                  if c in ("FETCH_SIZE", "WRITE_SIZE", "hbm_bytes_per_launch", "mfma_busy_frac")})


if __name__ == "__main__":
    main()
