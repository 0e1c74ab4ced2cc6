#!/usr/bin/env python3
"""Seeding 4 restarts of a large problem one at a time vs in one group (development measurement):
N = 2M, D = 128, 8 local trials (K = 1024), the first 33 centres."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "modular-trajectory-prediction-toolkit_amd"))
sys.path.insert(0, ROOT)
from bench import make_encodings  # noqa: E402
from pccs_accel import kernels as K  # noqa: E402
from pccs_accel import seeding  # noqa: E402

dev = torch.device("cuda:0")
n, d = 2_000_000, 128
X = make_encodings(n, d, 1024, seed=1000, device=dev)
K.sub_rowvec_(X, (K.col_sums(X) / n).to(torch.float32))
rs = np.random.RandomState(0)
draws = [(c, np.ascontiguousarray(u[:32])) for c, u in (seeding.draw(rs, n, 1024) for _ in range(4))]  # 8 trials, 33 centres
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    one = [seeding.kmeans_plusplus_batched(X, 33, [dr]) for dr in draws]
    torch.cuda.synchronize(); t1 = time.perf_counter()
    grp = seeding.kmeans_plusplus_batched(X, 33, draws)
    torch.cuda.synchronize(); t2 = time.perf_counter()
same = all(torch.equal(one[r][1][0], grp[1][r]) for r in range(4))
print(f"one at a time {1e3 * (t1 - t0):.1f} ms, one group {1e3 * (t2 - t1):.1f} ms, same seeds: {same}")
