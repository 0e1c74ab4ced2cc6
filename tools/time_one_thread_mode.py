#!/usr/bin/env python3
"""Wall time of the seeded eth-sized fit (N = 73 568, D = 96, K = 200, n_init = 20) in the default mode and in the
bit-for-bit sklearn-on-one-thread mode (sums="sklearn_1t").  python tools/time_one_thread_mode.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "modular-trajectory-prediction-toolkit_amd"), os.path.join(ROOT, "tests", "golden")]
import datagen  # noqa: E402
from pccs_accel import clustering  # noqa: E402

for name, seed, n, d, k_true, K in (("eth", 141, 73568, 96, 120, 200), ("sdd", 142, 17970, 96, 150, 400)):
    X = torch.from_numpy(datagen.make_encodings(seed, n, d, k_true, 1800, 420)).cuda()
    for mode in ("f64", "sklearn_1t"):
        best = None
        for rep in range(4):
            np.random.seed(7)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fit = clustering.kmeans_fit(X, K, n_init=20, sums=mode)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None or rep > 0 and dt < best else best
        print(f"{name} N={n} K={K} sums={mode}: {best * 1e3:.1f} ms (best of 3 after a warm-up), n_iter {fit['n_iter']}, inertia {fit['inertia']:.6f}", flush=True)
