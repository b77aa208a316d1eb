#!/usr/bin/env python3
"""Experiment: the inverse of two blocks at once (two threads, a context each) against one after the other: wall time per block and the
device time each block saw.   usage: two_in_flight_inverse.py [MiB]"""
import sys, time, json, os, threading
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dark-archon_amd"))
import torch
import pyarchon

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = mib << 20
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
x = torch.randint(0, 256, (n,), dtype=torch.uint8, device=dev, generator=g)
bwt = torch.empty(n, dtype=torch.uint8, device=dev)
base = torch.empty(1, dtype=torch.int32, device=dev)
pyarchon.forward_dev(x, None, bwt, base)
b0 = int(base.item())
outs = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(2)]
res = {}

def worker(t, reps, key):
    torch.cuda.set_device(0)
    pyarchon.bind_context(t, 0)
    st = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(st):
        ms = []
        for _ in range(reps):
            pyarchon.inverse_dev(bwt, b0, outs[t])
            ms.append(round(pyarchon.stats(0)["ms_total"], 3))
    res[(key, t)] = ms

for key, threads in (("alone", 1), ("two", 2), ("alone_again", 1)):
    th = [threading.Thread(target=worker, args=(t, 2, "warm")) for t in range(threads)]
    [t.start() for t in th]; [t.join() for t in th]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=worker, args=(t, 6, key)) for t in range(threads)]
    [t.start() for t in th]; [t.join() for t in th]
    torch.cuda.synchronize()
    res[(key, "wall_ms_per_block")] = round((time.perf_counter() - t0) * 1e3 / (6 * threads), 3)
assert bool(torch.equal(outs[0], x)) and bool(torch.equal(outs[1], x))
print(json.dumps({"%s/%s" % k: v for k, v in res.items() if k[0] != "warm"}))
