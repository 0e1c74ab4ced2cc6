#!/usr/bin/env python3
"""Development timing of the reference-sized configs (BASELINE.json configs[0..2]) end to end:
cluster_data (k-means++ x n_init restarts + Lloyd) + get_modality + best-of-20 eval on the GPU,
next to scikit-learn's KMeans on this box's CPU.  Not a bench line; numbers go to DESIGN.md."""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "modular-trajectory-prediction-toolkit_amd"))
sys.path.insert(0, ROOT)
from bench import make_encodings, make_trajectories  # noqa: E402

CONFIGS = {"eth": (73568, 96, 200, 2614), "hotel": (76646, 96, 200, 1075), "univ": (30128, 96, 200, 24334),
           "zara1": (74084, 96, 200, 2356), "zara2": (66976, 96, 200, 5910), "sdd": (17970, 96, 400, 2829)}

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="eth,sdd")
    ap.add_argument("--n-init", type=int, default=20)
    ap.add_argument("--cpu", action="store_true")
    ap.add_argument("--cpu-n-init", type=int, default=4, help="restarts of the sklearn run (time is scaled to --n-init)")
    ap.add_argument("--cpu-threads", default="", help="comma list of thread counts for extra sklearn runs "
                    "(the reference pins OMP_NUM_THREADS=8, PCCSNet/utils/utility.py:8)")
    args = ap.parse_args()
    from pccs_accel import clustering, seeding
    from pccs_accel import kernels as K
    from pccs_accel.evaluation import best_of_k
    dev = torch.device("cuda:0")
    for name in args.configs.split(","):
        n, d, k, nval = CONFIGS[name]
        X = make_encodings(n, d, k, seed=10, device=dev)
        Xh = X.cpu().numpy()
        for rep in range(2):
            np.random.seed(0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fit = clustering.kmeans_fit(X, k, n_init=args.n_init)
            torch.cuda.synchronize()
            t_fit = time.perf_counter() - t0
        # split: seeding alone
        Xc = X - X.mean(0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            seeding.kmeans_plusplus(Xc, k, np.random.RandomState(1))
        torch.cuda.synchronize()
        t_seed = (time.perf_counter() - t0) / 3
        t0 = time.perf_counter()
        mod = clustering.get_modality(X, fit["labels"], k)
        torch.cuda.synchronize()
        t_mod = time.perf_counter() - t0
        pred, gt = make_trajectories(nval, 20, 12, seed=4, device=dev)
        best_of_k(pred, gt, mode=0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        best_of_k(pred, gt, mode=0)
        torch.cuda.synchronize()
        t_eval = time.perf_counter() - t0
        print(f"{name}: N={n} K={k} n_init={args.n_init}: GPU fit {t_fit:.3f} s (k-means++ {t_seed*1e3:.1f} ms/init, "
              f"best n_iter {fit['n_iter']}, inertia {fit['inertia']:.2f}), get_modality {t_mod*1e3:.2f} ms, "
              f"best-of-20 eval ({nval} agents) {t_eval*1e3:.3f} ms", flush=True)
        if args.cpu:
            from sklearn.cluster import KMeans
            np.random.seed(0)
            t0 = time.perf_counter()
            km = KMeans(n_clusters=k, n_init=args.cpu_n_init).fit(Xh)
            t_cpu = (time.perf_counter() - t0) * args.n_init / args.cpu_n_init
            print(f"   sklearn CPU fit {t_cpu:.2f} s for n_init={args.n_init} (measured on {args.cpu_n_init} restarts, scaled; "
                  f"inertia {km.inertia_:.2f}, n_iter {km.n_iter_}) -> speedup {t_cpu/t_fit:.1f}x", flush=True)
            for nt in [int(x) for x in args.cpu_threads.split(",") if x]:
                from threadpoolctl import threadpool_limits
                with threadpool_limits(limits=nt):
                    np.random.seed(0)
                    t0 = time.perf_counter()
                    KMeans(n_clusters=k, n_init=args.cpu_n_init).fit(Xh)
                    t_nt = (time.perf_counter() - t0) * args.n_init / args.cpu_n_init
                print(f"   sklearn CPU fit at {nt} threads: {t_nt:.2f} s for n_init={args.n_init} -> speedup {t_nt/t_fit:.1f}x",
                      flush=True)
