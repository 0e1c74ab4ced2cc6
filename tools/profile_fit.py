#!/usr/bin/env python3
"""Where a reference-sized kmeans_fit spends its time (development aid)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "modular-trajectory-prediction-toolkit_amd"))
sys.path.insert(0, ROOT)
from bench import make_encodings  # noqa: E402
from pccs_accel import clustering, seeding  # noqa: E402
from pccs_accel import kernels as K  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "eth"
n, d, k = {"eth": (73568, 96, 200), "zara1": (74084, 96, 200), "sdd": (17970, 96, 400)}[name]
dev = torch.device("cuda:0")
X = make_encodings(n, d, k, seed=10, device=dev)


def sync():
    torch.cuda.synchronize()
    return time.perf_counter()


for rep in range(2):
    rs = np.random.RandomState(0)
    t0 = sync()
    tol_abs, mean32 = clustering.tolerance_and_mean(X, 1e-4)
    Xc = X.clone()
    K.sub_rowvec_(Xc, mean32)
    t1 = sync()
    draws = [seeding.draw(rs, n, k) for _ in range(20)]
    t2 = sync()
    c0s, _ = seeding.kmeans_plusplus_batched(Xc, k, draws)
    t3 = sync()
    labels, inertia, centers, n_iter = K.lloyd_batched(Xc, c0s, tol_abs=tol_abs, check_every=4)
    t4 = sync()
    best = None
    inertia = inertia.cpu().numpy()
    for b in range(20):
        if best is None or (inertia[b] < inertia[best] and not clustering.is_same_clustering(labels[b], labels[best], k)):
            best = b
    t5 = sync()
    print(f"{name} rep{rep}: prep {1e3*(t1-t0):.2f} ms, draws {1e3*(t2-t1):.2f}, kpp {1e3*(t3-t2):.2f}, "
          f"lloyd {1e3*(t4-t3):.2f} (n_iter {n_iter}), select {1e3*(t5-t4):.2f}", flush=True)
