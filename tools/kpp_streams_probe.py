#!/usr/bin/env python3
"""Batched k-means++ with the restarts split into groups on HIP streams of their own (development probe): one host
thread enqueues every group's launches; the groups' latency-bound launches can then overlap on the device."""
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
for name, n, k in (("eth", 73568, 200), ("sdd", 17970, 400)):
    X = make_encodings(n, 96, k, seed=10, device=dev)
    K.sub_rowvec_(X, (K.col_sums(X) / n).to(torch.float32))
    rs = np.random.RandomState(0)
    draws = [seeding.draw(rs, n, k) for _ in range(20)]
    ref = None
    for groups in (1, 2, 4, 1, 2, 4):
        streams = [torch.cuda.Stream(device=dev) for _ in range(groups)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs = []
        cur = torch.cuda.current_stream(dev)
        per = (20 + groups - 1) // groups
        for g, st in enumerate(streams):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                outs.append(seeding.kmeans_plusplus_batched(X, k, draws[g * per:(g + 1) * per]))
        for st in streams:
            cur.wait_stream(st)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        idx = torch.cat([o[1] for o in outs])
        if ref is None:
            ref = idx
        print(f"{name}: {groups} group(s) {1e3 * dt:.2f} ms, same seeds: {bool(torch.equal(idx, ref))}", flush=True)
