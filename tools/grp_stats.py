"""Tied-group sizes after D key bytes (CPU, numpy): where the rows of a block sit by group length.
Usage: python tools/grp_stats.py <MiB> <shape>   (256 MiB of prose: 6 minutes, 20 GB of host memory)"""
import sys, time
sys.path.insert(0, "dark-archon_amd")
import numpy as np, archon_synth as S
n = int(sys.argv[1]) << 20
shape = sys.argv[2]
t=time.time()
x = S.gen_shape(shape, n)
print("gen", time.time()-t, flush=True)
# a7 key of item s: x[s-1], x[s-2], ... ; group by first D bytes: D-gram ending at s-1 reversed. group sizes = counts of D-grams (as reversed strings) -> same multiset as forward D-grams
for D in (7, 14):
    t=time.time()
    m = n - D + 1
    if D == 7:
        k = np.zeros(m, dtype=np.uint64)
        for d in range(D):
            k = (k << np.uint64(8)) | x[d:d+m].astype(np.uint64)
        k.sort()
        b = np.flatnonzero(np.concatenate(([True], k[1:] != k[:-1], [True])))
    else:
        # two-level: sort by (first 7, next 7) via lexsort of two u64
        k1 = np.zeros(m, dtype=np.uint64); k2 = np.zeros(m, dtype=np.uint64)
        for d in range(7):
            k1 = (k1 << np.uint64(8)) | x[d:d+m].astype(np.uint64)
            k2 = (k2 << np.uint64(8)) | x[d+7:d+7+m].astype(np.uint64)
        o = np.lexsort((k2, k1))
        k1 = k1[o]; k2 = k2[o]
        b = np.flatnonzero(np.concatenate(([True], (k1[1:] != k1[:-1]) | (k2[1:] != k2[:-1]), [True])))
    sz = np.diff(b)
    tot = sz.sum()
    print("D", D, "groups", len(sz), "sort", time.time()-t, flush=True)
    edges = [1,2,3,5,9,17,33,65,129,257,513,1025,2049,4097,8193,16385,32769,65537,1<<18,1<<20,1<<22,1<<30]
    for lo,hi in zip(edges[:-1], edges[1:]):
        sel = (sz>=lo)&(sz<hi)
        print("  size [%d,%d): groups %d items %d (%.3f)" % (lo,hi,sel.sum(), sz[sel].sum(), sz[sel].sum()/tot))
