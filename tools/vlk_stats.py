"""Go / no-go measurement for a first stage on variable-length order-preserving keys (the reference's
Huffman-coded sort keys: bwt/a6/src/coder.c:81-100, bwt/a6/src/bwt.c:82-84,130-160 -- for a7 order the code has to be
ALPHABETIC, i.e. order-preserving and prefix-free, which plain Huffman is not).

CPU, numpy.  For a text-like block it reports the share of items that would land in 16-bit buckets above the in-LDS
sort's capacity (4608 records at 256 MiB; scaled with the sample) under
  bytes     the two plain key bytes the streaming stage buckets on today
  alpha     the first 16 bits of the concatenated codes of x[s-1], x[s-2], ... under the OPTIMAL alphabetic code of the
            block's byte counts (dynamic programme over the sorted alphabet)
  msd D     plain key bytes, buckets split again and again on the next byte until they fit: share of items still in an
            oversized bucket after D bytes (what a hybrid MSD split would have to carry to level D + 1)
Usage: python tools/vlk_stats.py <MiB> <shape|corpus> [...]        (64 MiB of text: about two minutes)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dark-archon_amd"))
import numpy as np  # noqa: E402
import archon_synth as S  # noqa: E402


def corpus(cap):
    exts = (".py", ".h", ".hpp", ".txt", ".md", ".rst", ".c", ".cpp", ".json", ".html", ".js", ".cmake")
    buf, seen = bytearray(), set()
    for top in ("/usr/lib/python3", "/usr/lib/python3.10", "/usr/local/lib/python3.10/dist-packages", "/opt/rocm/include", "/usr/include", "/usr/share"):
        for dp, dn, fn in os.walk(top):
            for f in sorted(fn):
                p = os.path.join(dp, f)
                try:
                    st = os.stat(p)
                    if not f.endswith(exts) or (st.st_size, f) in seen or st.st_size > (8 << 20) or os.path.islink(p):
                        continue
                    seen.add((st.st_size, f))
                    with open(p, "rb") as fh:
                        buf += fh.read()
                except OSError:
                    continue
                if len(buf) >= cap:
                    return np.frombuffer(bytes(buf[:cap]), np.uint8)
    return np.frombuffer(bytes(buf), np.uint8)


def alphabetic_code(counts):
    """optimal alphabetic (order-preserving prefix) code lengths for the symbols with count > 0: the classic O(m^3 / vector)
    dynamic programme cost[i][j] = min_k cost[i][k] + cost[k+1][j] + weight(i..j); returns (codes, lengths) per byte value"""
    syms = np.flatnonzero(counts)
    m = syms.size
    w = counts[syms].astype(np.float64)
    pre = np.concatenate(([0.0], np.cumsum(w)))
    cost = np.zeros((m, m))
    root = np.zeros((m, m), dtype=np.int32)
    for span in range(1, m):
        for i in range(m - span):
            j = i + span
            ks = np.arange(i, j)
            c = cost[i, ks] + cost[ks + 1, j]
            k = int(np.argmin(c))
            cost[i, j] = c[k] + pre[j + 1] - pre[i]
            root[i, j] = i + k
    codes = np.zeros(256, dtype=np.uint32)
    lens = np.zeros(256, dtype=np.uint32)
    stack = [(0, m - 1, 0, 0)]
    while stack:
        i, j, code, ln = stack.pop()
        if i == j:
            codes[syms[i]], lens[syms[i]] = code, max(ln, 1)
            continue
        k = root[i, j]
        stack.append((i, k, code << 1, ln + 1))
        stack.append((k + 1, j, (code << 1) | 1, ln + 1))
    return codes, lens, cost[0, m - 1] / w.sum() if m > 1 else 1.0


def share_over(keys, nbins, cap):
    h = np.bincount(keys, minlength=nbins)
    return float(h[h > cap].sum()) / keys.size, int((h > cap).sum()), int(h.max())


def main():
    mib = int(sys.argv[1])
    n = mib << 20
    cap = max(1, int(4608 * (n / (256 << 20))))
    for name in sys.argv[2:]:
        t0 = time.time()
        x = corpus(n) if name == "corpus" else S.gen_shape(name, n)
        n_eff = x.size
        counts = np.bincount(x, minlength=256)
        sigma = int((counts > 0).sum())
        print("%s: %d bytes, %d distinct, cap %d (= 4608 at 256 MiB)" % (name, n_eff, sigma, cap), flush=True)
        # plain bytes: bucket of item s = (x[s-1], x[s-2])
        k2 = (x[1:].astype(np.uint32) << 8) | x[:-1]
        sh, nb, mx = share_over(k2, 65536, cap)
        print("  bytes : %.3f of the items in %d oversized 16-bit buckets (largest %d)" % (sh, nb, mx), flush=True)
        # alphabetic code: first 16 bits of code(x[s-1]) code(x[s-2]) ...
        codes, lens, avg = alphabetic_code(counts)
        m = n_eff - 17
        acc = np.zeros(m, dtype=np.uint32)
        nbits = np.zeros(m, dtype=np.uint32)
        syms_per_key = np.zeros(m, dtype=np.uint8)
        for j in range(1, 17):
            sym = x[17 - j: 17 - j + m]                       # x[s-j] for s = 17 .. 17 + m - 1
            live = nbits < 16
            if not live.any():
                break
            c, l = codes[sym], lens[sym]
            acc = np.where(live, (acc << np.minimum(l, 16)) | c, acc)     # (bits beyond 16 are cut below)
            nb2 = nbits + l
            over = np.where(live & (nb2 > 16), nb2 - 16, 0)
            acc = np.where(live, acc >> over, acc)
            syms_per_key += live.astype(np.uint8)
            nbits = np.where(live, np.minimum(nb2, 16), nbits)
        acc = np.where(nbits < 16, acc << (16 - nbits), acc) & 0xFFFF
        sh, nb, mx = share_over(acc, 65536, cap)
        print("  alpha : %.3f of the items in %d oversized 16-bit buckets (largest %d); %.2f bits per symbol, %.2f symbols per 16-bit key"
              % (sh, nb, mx, avg, float(syms_per_key.mean())), flush=True)
        # hybrid MSD: D-gram buckets, D = 2 .. 7
        m = n_eff - 8
        key = np.zeros(m, dtype=np.uint64)
        for d in range(1, 8):
            key = (key << np.uint64(8)) | x[8 - d: 8 - d + m].astype(np.uint64)       # x[s-1] ... x[s-d] on top
            if d < 2:
                continue
            u, cnt = np.unique(key, return_counts=True)
            big = cnt > cap
            print("  msd %d : %.3f of the items in %d oversized buckets (largest %d)" % (d, float(cnt[big].sum()) / m, int(big.sum()), int(cnt.max())), flush=True)
            if not big.any():
                break
        print("  (%.0f s)" % (time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
