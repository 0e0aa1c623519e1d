#!/usr/bin/env python3
"""grad_bench.py -- the three run-sum gradient ops at bench_ops.py's shapes, alone (A/B of csrc/runsum.h variants:
EPNET_RUNSUM_QUAD=0 selects the row-at-a-time kernel). One JSON line per op and shape; every result is checked against a
float64 scatter-add of the same inputs first."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
from epnet_amd import pointnet2_cuda as p2, synth

dev = torch.device("cuda:0")
i32 = torch.int32
g = torch.Generator().manual_seed(0)


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


def report(op, shape, ms, nbytes, err):
    print(json.dumps({"op": op, "shape": shape, "ms": round(ms, 4), "GBps": round(nbytes / (ms * 1e-3) / 1e9, 1), "max_err": err,
                      "quad": os.environ.get("EPNET_RUNSUM_QUAD", "1")}), flush=True)


def f64_scatter(terms, flat, m):   # terms (b, c, p), flat (b, p) -> (b, c, m)
    out = torch.zeros((terms.shape[0], terms.shape[1], m), dtype=torch.float64, device=terms.device)
    out.scatter_add_(2, flat[:, None, :].expand(-1, terms.shape[1], -1), terms.double())
    return out


for bsz, (c, m, n) in ((16, (256, 4096, 16384)), (16, (512, 1024, 4096)), (1, (256, 4096, 16384)), (2, (128, 4096, 16384))):
    unknown = synth.scenes("kitti", bsz, n, seed=7).to(dev)
    kidx = torch.empty((bsz, m), dtype=i32, device=dev)
    known = torch.empty((bsz, m, 3), device=dev)
    p2.sample_centres_wrapper(bsz, n, m, unknown, p2.scene_index(unknown), kidx, known)
    d2 = torch.empty((bsz, n, 3), device=dev); idx = torch.empty((bsz, n, 3), dtype=i32, device=dev)
    p2.three_nn_wrapper(bsz, n, m, unknown, known, d2, idx)
    w = torch.rand((bsz, n, 3), generator=g).to(dev); w = (w / w.sum(-1, keepdim=True)).contiguous()
    go = torch.randn((bsz, c, n), generator=g).to(dev)
    gp = torch.zeros((bsz, c, m), device=dev)
    p2.three_interpolate_grad_wrapper(bsz, c, n, m, go, idx, w, gp)
    want = f64_scatter((go[:, :, :, None] * w[:, None, :, :]).reshape(bsz, c, n * 3), idx.reshape(bsz, -1).long(), m)
    err = float((gp.double() - want).abs().max())
    ms = timeit(lambda: p2.three_interpolate_grad_wrapper(bsz, c, n, m, go, idx, w, gp))
    report("three_interpolate_grad", {"B": bsz, "C": c, "n": n, "m": m}, ms, bsz * (c * n * 4 + n * 24 + c * m * 4), err)

for bsz, (c, n, m, ns, r) in ((16, (96, 4096, 1024, 32, 1.0)), (16, (96, 4096, 1024, 16, 0.5)), (16, (256, 1024, 256, 32, 2.0)), (2, (96, 4096, 1024, 32, 1.0))):
    pts = synth.scenes("kitti", bsz, n, seed=11).to(dev)
    cidx = torch.empty((bsz, m), dtype=i32, device=dev)
    ctr = torch.empty((bsz, m, 3), device=dev)
    p2.sample_centres_wrapper(bsz, n, m, pts, p2.scene_index(pts), cidx, ctr)
    idx = torch.empty((bsz, m, ns), dtype=i32, device=dev)
    p2.ball_query_wrapper(bsz, n, m, r, ns, ctr, pts, idx)
    go = torch.randn((bsz, c, m, ns), generator=g).to(dev)
    gp = torch.zeros((bsz, c, n), device=dev)
    p2.group_points_grad_wrapper(bsz, c, n, m, ns, go, idx, gp)
    want = f64_scatter(go.reshape(bsz, c, -1), idx.reshape(bsz, -1).long(), n)
    err = float((gp.double() - want).abs().max())
    ms = timeit(lambda: p2.group_points_grad_wrapper(bsz, c, n, m, ns, go, idx, gp))
    report("group_points_grad", {"B": bsz, "C": c, "N": n, "M": m, "ns": ns}, ms, bsz * (c * m * ns * 4 + m * ns * 4 + c * n * 4), err)
