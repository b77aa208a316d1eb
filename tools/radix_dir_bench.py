#!/usr/bin/env python3
"""Micro-benchmark mirroring tool/radix_dir/radix.c on the GPU.

The reference times four loop directions of a 256-bin counting-sort scatter over a 32 KiB source,
8192 repetitions (= 256 MiB scattered per figure; rez.*.txt: 0.89-1.75 s on 2006 CPUs).  Here the
same 256 MiB go through archon_hip_hist256 (the count, radix.c:31-36) and
archon_hip_radix_scatter (count + scan257 + run fill, radix.c:40-44) with the source in HBM.
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dark-archon_amd"))
import numpy as np
import torch
import pyarchon

SIZE, K = 1 << 15, 1 << 13
src32k = ((np.arange(SIZE, dtype=np.uint32) * 5423) & 0xFF).astype(np.uint8)       # radix.c:31-33
x = torch.from_numpy(np.tile(src32k, K)).cuda()                                     # 256 MiB
out = torch.zeros(256, dtype=torch.int32, device="cuda")
dst = torch.empty_like(x)


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


t_hist = timed(lambda: pyarchon.hist256_dev(x, out))
assert int(out.sum().item()) == x.numel()
t_scatter = timed(lambda: pyarchon.lib().archon_hip_radix_scatter_dev(x.data_ptr(), x.numel(), dst.data_ptr(), 0, None))
print(json.dumps({"bytes": x.numel(), "hist256_s": round(t_hist, 6), "hist256_GBps": round(x.numel() / t_hist / 1e9, 1),
                  "radix_scatter_s": round(t_scatter, 6), "radix_scatter_GBps": round(x.numel() / t_scatter / 1e9, 1),
                  "reference_cpu_s": {"celeron X++": 1.74, "core2duo X--": 0.89}}))
