#!/usr/bin/env python3
"""Development timing: the batched k-means++ (20 restarts) at the eth and SDD sizes, median of 7."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "modular-trajectory-prediction-toolkit_amd"), ROOT]
from bench import make_encodings  # noqa: E402
from pccs_accel import seeding  # noqa: E402

dev = torch.device("cuda:0")
for n, k in ((73568, 200), (17970, 400)):
    X = make_encodings(n, 96, k, seed=10, device=dev)
    X -= X.mean(0)
    rs = np.random.RandomState(1)
    draws = [seeding.draw(rs, n, k) for _ in range(20)]
    for _ in range(2):
        seeding.kmeans_plusplus_batched(X, k, draws)
    ts = []
    for _ in range(7):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        seeding.kmeans_plusplus_batched(X, k, draws)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    print(f"n={n} k={k} seeding_ms median {np.median(ts):.2f} min {min(ts):.2f}", flush=True)
