"""Small blocks (the container's default is 4 MiB, bwt/final/x3/archon.c:100): forward and inverse device time per block when
blocks are run one at a time, and the per-byte rate of a batch call (archon_hip_*_batch_dev) with 1 .. 8 workers.
Usage: python tools/small_blocks.py [MiB ...]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dark-archon_amd"))
import numpy as np, torch
import archon_synth as S, pyarchon
sizes = [int(a) for a in sys.argv[1:]] or [4, 16, 64]
for mib in sizes:
    n = mib << 20
    count = max(8, min(64, 256 // mib))
    xs = [torch.from_numpy(S.gen_shape("random", n, block=i)).cuda() for i in range(count)]
    bw = [torch.empty_like(t) for t in xs]
    bs = [torch.zeros(1, dtype=torch.int32, device="cuda") for _ in xs]
    outs = [torch.empty_like(t) for t in xs]
    # one at a time: device time from the library's own events, wall per block
    pyarchon.forward_dev(xs[0], None, bw[0], bs[0])
    t0 = time.perf_counter()
    dev_ms = []
    for i in range(count):
        pyarchon.forward_dev(xs[i], None, bw[i], bs[i])
        dev_ms.append(pyarchon.stats()["ms_total"])
    wall_f = (time.perf_counter() - t0) / count * 1e3
    bases = [int(b.item()) for b in bs]
    pyarchon.inverse_dev(bw[0], bases[0], outs[0])
    t0 = time.perf_counter()
    inv_ms = []
    for i in range(count):
        pyarchon.inverse_dev(bw[i], bases[i], outs[i])
        inv_ms.append(pyarchon.stats()["ms_total"])
    wall_i = (time.perf_counter() - t0) / count * 1e3
    row = {"block_MiB": mib, "blocks": count, "single_forward_device_ms": round(float(np.median(dev_ms)), 3), "single_forward_wall_ms": round(wall_f, 3),
           "single_inverse_device_ms": round(float(np.median(inv_ms)), 3), "single_inverse_wall_ms": round(wall_i, 3), "batch": []}
    for w in (1, 2, 4, 8):
        pyarchon.forward_batch_dev(xs, bw, bs, workers=w)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pyarchon.forward_batch_dev(xs, bw, bs, workers=w)
        torch.cuda.synchronize()
        tf = (time.perf_counter() - t0) / count * 1e3
        pyarchon.inverse_batch_dev(bw, bases, outs, workers=w)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pyarchon.inverse_batch_dev(bw, bases, outs, workers=w)
        torch.cuda.synchronize()
        ti = (time.perf_counter() - t0) / count * 1e3
        row["batch"].append({"workers": w, "forward_ms_per_block": round(tf, 3), "forward_GBps": round(n / tf / 1e6, 2),
                             "inverse_ms_per_block": round(ti, 3), "inverse_GBps": round(n / ti / 1e6, 2)})
    for i in range(count):
        assert torch.equal(outs[i], xs[i])
    print(json.dumps(row), flush=True)
