#!/usr/bin/env python3
"""Development timing: the batched Lloyd runs of a reference-sized fit (20 restarts from k-means++ seeds) for every
restart-group size of the E-step (MTP_RESTART_GROUP; 0 = one centroid image per restart).  Also checks that the
labels do not depend on the grouping."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "modular-trajectory-prediction-toolkit_amd"), ROOT]
from bench import make_encodings  # noqa: E402
from pccs_accel import kernels as K, seeding, clustering  # noqa: E402

dev = torch.device("cuda:0")
for name, n, k in (("eth", 73568, 200), ("univ", 30128, 200), ("sdd", 17970, 400)):
    X = make_encodings(n, 96, k, seed=10, device=dev)
    tol, mean = clustering.tolerance_and_mean(X, 1e-4)
    K.sub_rowvec_(X, mean)
    rs = np.random.RandomState(10)
    draws = [seeding.draw(rs, n, k) for _ in range(20)]
    c0s, _ = seeding.kmeans_plusplus_batched(X, k, draws)
    ref = None
    for group in ("0", "2", "3", "4", "5", "6", "8", ""):
        if group:
            os.environ["MTP_RESTART_GROUP"] = group
        else:
            os.environ.pop("MTP_RESTART_GROUP", None)
        out = K.lloyd_batched(X, c0s, tol_abs=tol, check_every=4)
        ts = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = K.lloyd_batched(X, c0s, tol_abs=tol, check_every=4)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        if ref is None:
            ref = out
        same = torch.equal(out[0], ref[0]) and out[3] == ref[3]
        print(f"{name} N={n} K={k} group={group or 'auto'}: lloyd_ms median {np.median(ts):.2f} min {min(ts):.2f} "
              f"iters(sum)={sum(out[3])} max={max(out[3])} same_labels={same}", flush=True)
