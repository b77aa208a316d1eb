#!/usr/bin/env python3
"""Stress for several blocks in flight on one GPU: F feeder threads (a compute context each) run blocks of different shapes and sizes in
different orders, round after round; EVERY output (SA, BWT, primary index) is compared on the device with the result the same block gave
alone.  usage: stress_in_flight.py [rounds] [feeders] [MiB of the big blocks]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dark-archon_amd"))
import numpy as np
import torch
import archon_synth
import pyarchon

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
F = int(sys.argv[2]) if len(sys.argv) > 2 else 2
big = (int(sys.argv[3]) if len(sys.argv) > 3 else 256) << 20
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
shapes = [("random", big), ("dna", big), ("a", big), ("text", big // 4), ("random", big // 2 + 12345), ("motif", big // 2), ("dna", (16 << 20) + 7),
          ("random", 4 << 20), ("prose", big // 8)]
blocks, refs = [], []
for name, n in shapes:
    x = torch.from_numpy(archon_synth.gen_shape(name, n)).to(dev)
    sa = torch.empty(n, dtype=torch.int32, device=dev)
    bwt = torch.empty(n, dtype=torch.uint8, device=dev)
    base = torch.empty(1, dtype=torch.int32, device=dev)
    pyarchon.forward_dev(x, sa, bwt, base)
    assert pyarchon.validate_dev(x, sa), name
    blocks.append(x)
    refs.append((sa, bwt, int(base.item())))
torch.cuda.synchronize()
bad = []
done = [0] * F


def feeder(t):
    torch.cuda.set_device(0)
    pyarchon.bind_context(t, 0)
    st = torch.cuda.Stream(device=dev)
    nmax = max(b.numel() for b in blocks)
    sa = torch.empty(nmax, dtype=torch.int32, device=dev)
    bwt = torch.empty(nmax, dtype=torch.uint8, device=dev)
    base = torch.empty(1, dtype=torch.int32, device=dev)
    rng = np.random.default_rng(100 + t)
    with torch.cuda.stream(st):
        for r in range(rounds):
            for i in rng.permutation(len(blocks)):
                x = blocks[i]
                n = x.numel()
                pyarchon.forward_dev(x, sa[:n], bwt[:n], base)
                ok = int(base.item()) == refs[i][2] and bool(torch.equal(bwt[:n], refs[i][1])) and bool(torch.equal(sa[:n], refs[i][0]))
                if not ok:
                    bad.append((t, r, shapes[i], pyarchon.stats(0)))
                done[t] += 1


t0 = time.time()
ts = [threading.Thread(target=feeder, args=(t,)) for t in range(F)]
for t in ts:
    t.start()
for t in ts:
    t.join()
torch.cuda.synchronize()
print("feeders %d rounds %d blocks %d seconds %.1f failures %d" % (F, rounds, sum(done), time.time() - t0, len(bad)))
for b in bad[:5]:
    print(b)
sys.exit(1 if bad else 0)
