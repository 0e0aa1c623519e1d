"""CPU simulation of the bucket pruning of furthest point sampling (design study of round 1: how many buckets a round touches for a
given bucket size). Test-side tooling: it takes the sample sequence from the oracle, so it lives under tests/ (only tests, smoke() and
bench.py's baseline / verification legs use oracle/)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from epnet_amd import synth
from oracle import oracle

def morton_order(p, bits=10):
    lo, hi = p.min(0), p.max(0)
    q = ((p - lo) / np.maximum(hi - lo, 1e-9) * (2**bits - 1)).astype(np.uint32)
    code = np.zeros(len(p), np.uint64)
    for b in range(bits):
        for d in range(3):
            code |= ((q[:, d] >> b) & 1).astype(np.uint64) << np.uint64(3 * b + d)
    return np.argsort(code, kind="stable")

def sim(kind, n, m, bsize, seed=1):
    p = synth.scenes(kind, 1, n, seed=seed).numpy()[0]
    order = morton_order(p)
    ps = p[order]
    nb = n // bsize
    bmin = ps.reshape(nb, bsize, 3).min(1); bmax_ = ps.reshape(nb, bsize, 3).max(1)
    idx = oracle.furthest_point_sampling(p[None], m)[0]
    t = np.full(n, 1e10, np.float32)
    tb = t.reshape(nb, bsize)
    active_counts = []
    for it in range(1, m):
        c = p[idx[it - 1]]
        cl = np.clip(c, bmin, bmax_)
        L = ((cl - c).astype(np.float32) ** 2).sum(1)
        bm = tb.max(1)
        act = L < bm
        active_counts.append(act.sum())
        d = ((ps - c) ** 2).sum(1).astype(np.float32)
        # verify skipping is exact
        newt = np.minimum(t, d)
        changed = (newt != t).reshape(nb, bsize).any(1)
        assert not (changed & ~act).any()
        t[:] = newt
    a = np.array(active_counts)
    print("%s n=%d m=%d bucket=%d (%d buckets): active/iter mean %.1f median %.0f p90 %.0f max %d; first 64 iters mean %.1f" % (kind, n, m, bsize, nb, a.mean(), np.median(a), np.percentile(a, 90), a.max(), a[:64].mean()))

for kind in ("kitti", "ubox"):
    sim(kind, 16384, 4096, 64)
    sim(kind, 16384, 4096, 32)
    sim(kind, 4096, 1024, 64)
    sim(kind, 4096, 1024, 16)
