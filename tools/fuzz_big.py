"""ad-hoc bug hunt at sizes the CPU oracle is too slow for: the natural-route generators of tests/test_gpu_fuzz.py scaled
to 16-80 MiB, checked by LF-consistency of the SA on the device + inverse round trip (size-independent properties)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dark-archon_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, pyarchon
import test_gpu_fuzz as T

def big_cases(rng, count):
    for _ in range(count):
        n = int(rng.choice([1 << 24, (1 << 24) + 1, 20000003, 1 << 25, 50331649, 80000000]))
        kind = int(rng.integers(0, 7))
        if kind == 0:
            x = rng.integers(0, 256, size=n, dtype=np.uint8)
        elif kind == 1:
            k = int(rng.choice([2, 3, 4, 5, 12, 16, 17, 40, 200]))
            x = rng.choice(np.sort(rng.choice(256, size=k, replace=False)).astype(np.uint8), size=n)
            if rng.integers(0, 2): x[int(rng.integers(n // 2, n))] = np.uint8(rng.integers(0, 256))
        elif kind == 2:
            m = int(rng.choice([1, 2, 3, 7, 100, 1000, 4099, 65537]))
            x = np.tile(rng.integers(0, 256, size=m, dtype=np.uint8), n // m + 1)[:n].copy()
            for _ in range(int(rng.integers(0, 4))): x[int(rng.integers(0, n))] ^= np.uint8(1 + rng.integers(0, 255))
        elif kind == 3:
            x = rng.integers(97, 123, size=n, dtype=np.uint8)
            L = int(rng.integers(10, n // 3)); a, b = int(rng.integers(0, n - L)), int(rng.integers(0, n - L))
            x[b:b + L] = x[a:a + L].copy()
        elif kind == 4:
            vals = rng.integers(0, 256, size=n // 20 + 2, dtype=np.uint8)
            x = np.repeat(vals, rng.integers(1, 40, size=vals.size))[:n]
            if x.size < n: x = np.concatenate([x, rng.integers(0, 256, size=n - x.size, dtype=np.uint8)])
        elif kind == 5:
            h = n // 2
            x = np.concatenate([rng.choice(np.frombuffer(b"ACGT", np.uint8), size=h), rng.integers(0, 256, size=n - h, dtype=np.uint8)])
        else:
            x = np.where(rng.random(n) < 0.9, 255, rng.integers(250, 256, size=n)).astype(np.uint8)
        yield kind, np.ascontiguousarray(x, np.uint8)

bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(seed)
    for kind, x in big_cases(rng, 6):
        n = x.size
        x_t = torch.from_numpy(x).cuda()
        sa_t = torch.empty(n, dtype=torch.int32, device="cuda"); bwt_t = torch.empty(n, dtype=torch.uint8, device="cuda")
        base_t = torch.zeros(1, dtype=torch.int32, device="cuda"); out_t = torch.empty(n, dtype=torch.uint8, device="cuda")
        try:
            pyarchon.forward_dev(x_t, sa_t, bwt_t, base_t)
            st = pyarchon.stats()
            ok = pyarchon.validate_dev(x_t, sa_t)
            pyarchon.inverse_dev(bwt_t, int(base_t.item()), out_t)
            ok = ok and bool(torch.equal(out_t, x_t))
        except Exception as e:
            ok = False; st = {}; print("EXC", e)
        print("seed", seed, "kind", kind, "n", n, "ok", ok, {k: st.get(k) for k in ("path", "alphabet_bits", "doubling_rounds", "period", "ms_total")}, flush=True)
        bad += 0 if ok else 1
        del x_t, sa_t, bwt_t, out_t
print("failures:", bad)
