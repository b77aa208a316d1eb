#!/usr/bin/env python3
"""Experiment: K graded blocks (256 MiB uniform random, SA + BWT out) one after the other on one context, against the same K blocks dealt to
2 / 3 worker threads with a context each (archon_hip_forward_batch_dev): does a second block in flight fill the tails of the first one's
kernels and the host's round trip?   usage: two_in_flight.py [MiB] [K]"""
import sys, time, json, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dark-archon_amd"))
import numpy as np
import torch
import pyarchon

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n = mib << 20
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
xs = [torch.randint(0, 256, (n,), dtype=torch.uint8, device=dev, generator=g) for _ in range(2)]
x_ts = [xs[i & 1] for i in range(K)]
sa_ts = [torch.empty(n, dtype=torch.int32, device=dev) for _ in range(K)]
bwt_ts = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(K)]
base_ts = [torch.empty(1, dtype=torch.int32, device=dev) for _ in range(K)]
torch.cuda.synchronize()
out = {"MiB": mib, "K": K}
for _ in range(3):
    pyarchon.forward_dev(x_ts[0], sa_ts[0], bwt_ts[0], base_ts[0])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(K):
    pyarchon.forward_dev(x_ts[i], sa_ts[i], bwt_ts[i], base_ts[i])
torch.cuda.synchronize()
out["one_context_ms_per_block"] = round((time.perf_counter() - t0) * 1e3 / K, 4)
ref = [(bwt_ts[i][:4096].clone(), int(base_ts[i].item())) for i in range(2)]
for w in (1, 2, 3, 4):
    pyarchon.forward_batch_dev(x_ts[:2 * w], bwt_ts[:2 * w], base_ts[:2 * w], sa_ts[:2 * w], workers=w)      # warm the workers' contexts
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        pyarchon.forward_batch_dev(x_ts, bwt_ts, base_ts, sa_ts, workers=w)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) * 1e3 / K)
    for i in range(K):
        assert int(base_ts[i].item()) == ref[i & 1][1] and bool((bwt_ts[i][:4096] == ref[i & 1][0]).all())
    out["workers_%d_ms_per_block" % w] = round(best, 4)
    out["workers_%d_GBps" % w] = round(n / best / 1e6, 2)
print(json.dumps(out))
