#!/usr/bin/env python3
"""bench.py -- SA-stack points/s per GPU on synthetic KITTI-shaped scenes (BASELINE.json's metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--pipelined 0|1] [--kind kitti|ubox|dup|kitti_q] [--no-graph]

One "step" = one pass of the 4-level set-abstraction OPERATOR stack (per level: scene index, sampling [FPS + centre gather; levels
2-4 nested in level 1 where tie-free], two ball queries, two fused groupings -- epnet_amd/sa_stack.py) over a batch of B independent
16384-point scenes that are already resident in HBM (default B = 256: 12.5 GB of resident buffers out of 288 GB). Steps are
software-pipelined by default: the latency-bound sampling chain of step k runs beside the bandwidth-bound ball query + grouping of
step k-1 (double-buffered inputs / centres / indices); every step does the full work of one batch and consumes a DIFFERENT batch
than the step before (two resident batches alternate, as a loop that takes a new batch per iteration does:
tools/train_rcnn.py:221-223); --pipelined 0 runs every step on its own (the latency figure). N > 1: launched by
torch.distributed.run, one rank per GPU; every rank owns its own scenes (different seeds), there is NO collective in the data path
(the ops are independent per scene), so scaling is "weak". Timing: barrier + synchronize on both sides of exactly K steps, MAX over
ranks; rank 0 prints ONE JSON line.

Extra objects on that line:
  roofline      the op family with the largest share of the step (pick_dominant: among families within 10 % of the longest the
                one that moves the most bytes -- the level-1 sampling's chain of rounds and the grouping calls are 2.1 - 2.2 ms each
                and trade places from run to run; `dominant_by_time` names the longest of this run), priced by its ALGORITHMIC
                bytes per launch / its average duration measured with HIP events on the launch stream over an instrumented
                (unpipelined, single-stream) run of the same K steps; `kernels` lists the same figure for every op family,
                `roofline_hbm_bound` repeats it for the largest bandwidth-bound one (the grouping).
  cpu_baseline  the CPU oracle (a scalar C port of the reference kernels, 1 core) timed on this host on a bounded sample of the
                same workload.
  verified      scene 0 of the buffers the timed steps left behind, every tensor against the oracle, each resident batch in each
                role (verification.distinct_inputs).
  ms_per_step_min / _median / _max   per-step durations from an event pair around every step (a loop of its own after the timed one)
  devices_distinct, devices, per_rank_ms_per_step   what the ranks computed on (PCI address / UUID of every rank's device)
  sampling_chain, kinds, latency_one_scene, with_fp, config5   the chain's identity share per level; the same step on the other
                input families (uniform box, padded with exact twins, decimal-quantised); further verified measurements
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None,
                    help="scenes per GPU per step (default 256; EPNET_BENCH_BATCH overrides)")
    ap.add_argument("--config", type=int, default=2, choices=[2, 5],
                    help="BASELINE.json config: 2 = 16384-point scenes through the 4-level stack (the metric's configuration), "
                         "5 = dense 65536-point scenes, one level: FPS 16384, ball query r 0.5 / nsample 64, grouping C = 3 and 64")
    ap.add_argument("--points", type=int, default=None, help="points per scene (default: the config's)")
    ap.add_argument("--kind", default="kitti", choices=["kitti", "ubox", "dup", "kitti_q"])
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a HIP graph")
    ap.add_argument("--no-overlap", action="store_true",
                    help="single stream: do not run ball query / grouping of level l beside the FPS of level l+1")
    ap.add_argument("--pipelined", type=int, default=int(os.environ.get("EPNET_BENCH_PIPELINED", "1")),
                    help="1: software-pipelined steps -- the (latency-bound) sampling chain of step k runs beside the "
                         "(bandwidth-bound) ball query + grouping of step k-1, double-buffered; 0: every step alone")
    ap.add_argument("--stages", type=int, default=int(os.environ.get("EPNET_BENCH_STAGES", "2")), choices=[2, 3],
                    help="software pipeline depth: 2 = sampling chain of step k beside grouping of step k-1; 3 = level-1 sampling of "
                         "step k beside the rest of the sampling chain + ball queries of step k-1 beside the grouping of step k-2 (measured "
                         "slower: 3.47 against 3.24 ms -- the short kernels of the middle stage need 126 - 129 registers per lane and find no "
                         "place on a CU while the level-1 sampling holds 400 of its 512: profiles/r03_timeline_three_stages.txt)")
    ap.add_argument("--unfused", action="store_true", help="grouping as two group_points calls instead of group_concat")
    ap.add_argument("--no-shared-index", action="store_true", help="every op sorts the scene for itself")
    ap.add_argument("--fused-sampling", action="store_true", help="(default; kept for old command lines)")
    ap.add_argument("--module-sampling", action="store_true",
                    help="sample every level with the reference module's op-by-op sequence (transpose, fill temp, FPS, gather on the "
                         "flipped cloud, transpose back: pointnet2_modules.py:30-45) instead of epnet_sample_centres (FPS + row gather "
                         "of the centres, same idx / new_xyz): 6 instead of 3 launches per level on the sampling chain")
    ap.add_argument("--with-fp", action="store_true", help="also run the 4 three_nn + 4 three_interpolate FP ops")
    ap.add_argument("--cpu-scenes", type=int, default=None,
                    help="scenes in the cpu_baseline sample (0 = skip; default 16 for config 2, 2 for config 5)")
    ap.add_argument("--sweep", default="", help="comma list of extra batch sizes to time (reported under 'sweep')")
    ap.add_argument("--extras", default="b1,with_fp,kinds,config5",
                    help="further measurements reported as sub-objects of the default N = 1 line (comma list of b1, with_fp, kinds, config5; "
                         "empty = none)")
    ap.add_argument("--verify-scenes", type=int, default=1,
                    help="scenes of the timed buffers rank 0 checks against the CPU oracle after the timed loop (0 = skip; "
                         "the line then carries verified: null)")
    ap.add_argument("--rehearsal", action="store_true",
                    help="allow EPNET_BENCH_DEVICE / EPNET_BENCH_BACKEND (all ranks on one device, gloo instead of RCCL): the line "
                         "then says rehearsal: true and its devices_distinct shows it; without the flag both variables are refused")
    ap.add_argument("--launch-check", action="store_true",
                    help="rehearse the N-rank launch only: the ranks rendezvous (backend EPNET_BENCH_BACKEND, default nccl), "
                         "count themselves with an all-reduce and rank 0 prints n_gpus / ranks_seen; no kernels of the path")
    args = ap.parse_args()
    from epnet_amd import sa_stack
    args.cfg = sa_stack.CONFIGS[args.config]
    if args.points is None:
        args.points = args.cfg["n"]
    if args.batch is None:
        args.batch = int(os.environ.get("EPNET_BENCH_BATCH", "256"))   # (config 5: 80 GB of resident buffers; one sampling workgroup per CU)
    if args.cpu_scenes is None:
        args.cpu_scenes = int(os.environ.get("EPNET_BENCH_CPU_SCENES", "16" if args.config == 2 else "2"))
    if args.with_fp and args.config != 2:
        ap.error("--with-fp belongs to config 2 (the FP modules of the RPN pyramid)")
    return args


_CPU_JOBS = {}  # inputs of the CPU baselines, prepared in the parent and inherited by forked workers


def _cpu_prepare(kind, n_points, seeds, cfg):
    import numpy as np
    from epnet_amd import synth
    rng = np.random.default_rng(0)
    _CPU_JOBS["cfg"] = cfg
    _CPU_JOBS["feats"] = [None if c == 0 else rng.standard_normal((1, c, nn)).astype(np.float32)
                          for c, nn in zip(cfg["feat_channels"], (n_points,) + tuple(cfg["npoints"][:-1]))]
    for s in seeds:
        _CPU_JOBS[s] = synth.scenes(kind, 1, n_points, seed=s).numpy()


def _cpu_scene_seconds(seed):
    """the oracle's scalar port of the same op stack on one prepared scene; returns its CPU seconds"""
    import numpy as np
    from oracle import oracle
    feats, cur, cfg = _CPU_JOBS["feats"], _CPU_JOBS[seed], _CPU_JOBS["cfg"]
    t0 = time.perf_counter()
    for lvl, m in enumerate(cfg["npoints"]):
        cur_t = np.ascontiguousarray(cur.transpose(0, 2, 1))
        idx = oracle.furthest_point_sampling(cur, m)
        new_xyz = np.ascontiguousarray(oracle.gather_points(cur_t, idx).transpose(0, 2, 1))
        for radius, ns in zip(cfg["radii"][lvl], cfg["nsamples"][lvl]):
            bq = oracle.ball_query(radius, ns, cur, new_xyz)
            oracle.group_points(cur_t, bq)
            if feats[lvl] is not None:
                oracle.group_points(feats[lvl], bq)
        cur = new_xyz
    return time.perf_counter() - t0


def cpu_baseline(kind, n_points, scenes, cfg):
    """one core: `scenes` scenes one after the other"""
    from oracle import oracle
    oracle.build()
    seeds = [1000 + s for s in range(scenes)]
    _cpu_prepare(kind, n_points, seeds, cfg)
    t_total = sum(_cpu_scene_seconds(s) for s in seeds)
    return {"value": scenes * n_points / t_total, "unit": "points/s", "cores": 1, "kind": "port",
            "sample": "%d %s scenes of %d points through the same %d-level SA op stack, oracle/epnet_oracle.c, "
                      "%.1f s of CPU time on %d-core host" % (scenes, kind, n_points, len(cfg["npoints"]), t_total, os.cpu_count())}


def cpu_baseline_multicore(kind, n_points, procs, cfg, per_proc=2):
    """the same port, one scene per worker process at a time (the stack shards by scene on the CPU too); the
    workers are forked BEFORE this process touches the GPU and inherit the prepared inputs"""
    import multiprocessing as mp
    from oracle import oracle
    oracle.build()
    seeds = [2000 + s for s in range(per_proc * procs)]
    _cpu_prepare(kind, n_points, seeds, cfg)
    with mp.get_context("fork").Pool(procs) as pool:
        pool.map(_cpu_scene_seconds, seeds[:procs], chunksize=1)   # start-up: workers load the oracle
        t0 = time.perf_counter()
        pool.map(_cpu_scene_seconds, seeds, chunksize=1)
        wall = time.perf_counter() - t0
    return {"value": len(seeds) * n_points / wall, "unit": "points/s", "cores": procs, "kind": "port",
            "sample": "%d %s scenes over %d worker processes, %.1f s wall" % (len(seeds), kind, procs, wall)}


def verify_scene(stack, xyz, scene, prev_xyz=None, prev2_xyz=None):
    """One scene of the buffers the timed steps left behind, checked against the CPU oracle DIRECTLY (not against another
    HIP path): every level's FPS indices and centres, both ball-query index tensors, both grouped tensors
    [xyz - centre ; features] (and, with the FP ops in the step, three_nn / three_interpolate). Integer outputs and copies
    must be identical; the interpolation is held to 1e-5. Returns the list of mismatching outputs (empty = verified).

    A software-pipelined stack holds several batches after a step (SAStack.owners() / owners3()): `xyz` is the batch the last step
    sampled, `prev_xyz` the batch of the step before, `prev2_xyz` (three stages) the one before that; None: the same batch again.
      two stages    stage S buffers (fps_idx, the sampling parity's sets / stage-S ball queries / FP searches) <-> xyz;
                    stage G buffers (the other parity's sets, stage-G ball queries, grouped tensors, interpolation) <-> prev_xyz
      three stages  slot S1: level-1 fps_idx <-> xyz (nothing else of that slot is current); slot S2: everything the sampling
                    chain writes <-> prev_xyz; slot G: the same and every stage-G output <-> prev2_xyz (a complete batch)
    The oracle is the checker here, outside every timed region (oracle/oracle.py header)."""
    import numpy as np
    from oracle import oracle
    oracle.build()
    bad = []

    def same(name, got, want, tol=None):
        got = got.detach().cpu().numpy()
        ok = np.array_equal(got, want) if tol is None else np.allclose(got, want, rtol=tol, atol=tol)
        if not ok:
            bad.append(name)

    def expected(cloud):
        """what the reference's op sequence produces for one cloud, level by level"""
        cur = cloud[scene:scene + 1].detach().cpu().numpy()
        clouds, per_level = [cur], []
        for L in stack.levels:
            cur_t = np.ascontiguousarray(cur.transpose(0, 2, 1))
            fps = oracle.furthest_point_sampling(cur, L["m"])
            new_xyz = np.ascontiguousarray(oracle.gather_points(cur_t, fps).transpose(0, 2, 1))
            bqs = [oracle.ball_query(S["radius"], S["ns"], cur, new_xyz) for S in L["scales"]]
            per_level.append({"cur": cur, "cur_t": cur_t, "fps": fps, "new_xyz": new_xyz, "bq": bqs})
            cur = new_xyz
            clouds.append(cur)
        fp = []
        for k, F in enumerate(stack.fp_bufs if stack.with_fp else []):
            known, unknown = clouds[len(stack.levels) - k], clouds[len(stack.levels) - k - 1]
            d2, nn_idx = oracle.three_nn(unknown, known)
            fp.append((d2, nn_idx))
        return per_level, fp

    cache = []

    def want_of(cloud):
        for c, w in cache:
            if c is cloud:
                return w
        cache.append((cloud, expected(cloud)))
        return cache[-1][1]

    prev_xyz = xyz if prev_xyz is None else prev_xyz
    prev2_xyz = prev_xyz if prev2_xyz is None else prev2_xyz
    stages = getattr(stack, "stages", 2 if getattr(stack, "pipelined", False) else 1)
    # slot -> (expectation, "all" | "level-1 sampling only"); the batch stage G worked on
    if stages == 3:
        s1, s2, g = stack.owners3()
        by_slot = {s1: (want_of(xyz), "l1"), s2: (want_of(prev_xyz), "all"), g: (want_of(prev2_xyz), "all")}
        want_g = want_of(prev2_xyz)
    elif stages == 2:
        s_par, g_par = stack.owners()
        by_slot = {s_par: (want_of(xyz), "all"), g_par: (want_of(prev_xyz), "all")}
        want_g = want_of(prev_xyz)
    else:
        by_slot = {0: (want_of(xyz), "all")}
        want_g = want_of(xyz)

    for lvl, L in enumerate(stack.levels):
        tag = "level%d." % (lvl + 1)
        fps_sets = L.get("fps_idx_sets") or [L["fps_idx"]]
        if all(t is fps_sets[0] for t in fps_sets):      # one tensor: the sampling chain of the last step's batch wrote it
            same(tag + "fps_idx", fps_sets[0][scene:scene + 1], want_of(xyz)[0][lvl]["fps"])
        else:
            for k, t in enumerate(fps_sets):
                want, scope = by_slot[k]
                if scope == "all" or lvl == 0:
                    same(tag + "fps_idx[set %d]" % k, t[scene:scene + 1], want[0][lvl]["fps"])
        for k, P in enumerate(L["sets"]):
            want, scope = by_slot[k]
            if scope == "all":
                same(tag + "new_xyz[set %d]" % k, P["new_xyz"][scene:scene + 1], want[0][lvl]["new_xyz"])
        feats = None if L["features"] is None else L["features"][scene:scene + 1].cpu().numpy()
        G = want_g[0][lvl]
        for j, S in enumerate(L["scales"]):
            stag = tag + "r%g." % S["radius"]
            idx_sets = S.get("idx_sets", [S["idx"]])
            for k, idx_set in enumerate(idx_sets):   # one per ring slot when the queries run in stage S, else stage G's own
                if len(idx_sets) > 1:
                    want, scope = by_slot[k]
                    if scope != "all":
                        continue
                    want_bq = want[0][lvl]["bq"][j]
                else:
                    want_bq = G["bq"][j]
                same(stag + "ball_idx[set %d]" % k, idx_set[scene:scene + 1], want_bq)
            bq = G["bq"][j]
            want_xyz = oracle.group_points(G["cur_t"], bq) - G["new_xyz"].transpose(0, 2, 1)[:, :, :, None]   # pointnet2_utils.py:250-251
            want_feat = None if feats is None else oracle.group_points(feats, bq)
            if stack.fused:
                want = want_xyz if want_feat is None else np.concatenate([want_xyz, want_feat], axis=1)  # :254-257
                same(stag + "grouped", S["grouped"][scene:scene + 1], want)
            else:
                same(stag + "grouped_xyz", S["grouped_xyz"][scene:scene + 1], oracle.group_points(G["cur_t"], bq))
                if want_feat is not None:
                    same(stag + "grouped_feat", S["grouped_feat"][scene:scene + 1], want_feat)
    for k, F in enumerate(stack.fp_bufs if stack.with_fp else []):
        tag = "fp%d." % (len(stack.levels) - k)
        for q, P in enumerate(F["sets"]):           # the searches run in stage S (one set per ring slot)
            want, scope = by_slot[q] if len(F["sets"]) > 1 else (want_of(xyz), "all")
            if scope != "all":
                continue
            d2, nn_idx = want[1][k]
            same(tag + "three_nn.idx[set %d]" % q, P["idx"][scene:scene + 1], nn_idx)
            same(tag + "three_nn.dist2[set %d]" % q, P["dist2"][scene:scene + 1], d2)
        d2, nn_idx = want_g[1][k]                   # the interpolation runs in stage G
        inv = (np.float32(1.0) / (np.sqrt(d2) + np.float32(1e-8))).astype(np.float32)     # pointnet2_modules.py:157-159
        weight = (inv / inv.sum(axis=2, keepdims=True)).astype(np.float32)
        want = oracle.three_interpolate(F["known_feats"][scene:scene + 1].cpu().numpy(), nn_idx, weight)
        same(tag + "three_interpolate", F["out"][scene:scene + 1], want, tol=1e-5)
    return bad


class OpTimer:
    """wraps the extension stand-in's functions with HIP event pairs recorded on the launch stream"""

    def __init__(self, torch, ext, names=None):
        """names: the functions of `ext` to wrap (default: every *_wrapper of the pointnet2 stand-in)"""
        self.torch, self.ext, self.records, self.saved = torch, ext, [], {}
        self.names = names

    def __enter__(self):
        for name in (self.names if self.names is not None else [n for n in dir(self.ext) if n.endswith("_wrapper")]):
            fn = getattr(self.ext, name)
            self.saved[name] = fn

            def timed(*a, _fn=fn, _name=name, **kw):
                dev = next(x for x in a if hasattr(x, "is_cuda")).device
                stream = self.torch.cuda.current_stream(dev)
                e0 = self.torch.cuda.Event(enable_timing=True)
                e1 = self.torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                r = _fn(*a, **kw)
                e1.record(stream)
                self.records.append((_name, a[:5], e0, e1))
                return r

            setattr(self.ext, name, timed)
        return self

    def __exit__(self, *exc):
        for name, fn in self.saved.items():
            setattr(self.ext, name, fn)
        return False


def pick_dominant(kernels):
    """(longest, dominant) op family of a step. The dominant family is the longest per step -- among the families within 10 % of
    the longest (the level-1 sampling's 2.1 ms chain of rounds and the four grouping calls' 2.1 - 2.2 ms trade places from run to run)
    the one that moves the most algorithmic bytes, so that the line names the same kernel every run; `roofline.dominant_by_time`
    says which was longest in this one."""
    longest = max(kernels, key=lambda k: kernels[k]["step_ms"])
    near = [k for k in kernels if kernels[k]["step_ms"] >= 0.9 * kernels[longest]["step_ms"]]
    return longest, max(near, key=lambda k: kernels[k]["bytes_per_launch"] * kernels[k]["launches_per_step"])


def op_family(name, head):
    """(family label, algorithmic bytes of this launch) -- formulas of SURVEY.md section 8(d)"""
    if name.startswith("scene_index"):
        b, n = head[:2]
        np_ = 2048
        while np_ < n:
            np_ *= 2
        return "scene_index N=%d" % n, b * (n * 12 + np_ * 16 + np_ // 64 * 24)
    if name.startswith("furthest"):
        b, n, m = head[:3]
        return "fps N=%d M=%d" % (n, m), b * (n * 12 + m * 4)
    if name.startswith("sample_centres"):  # FPS + the gather of the centres: the bytes of both
        b, n, m = head[:3]
        return "fps N=%d M=%d" % (n, m), b * (n * 12 + m * 4) + b * (m * 4 + 3 * n * 4 + 3 * m * 4)
    if name.startswith("gather_points_w"):
        b, c, n, m = head[:4]
        return "gather", b * (m * 4 + c * n * 4 + c * m * 4)
    if name.startswith(("ball_query_multi", "ball_query_ordered")):  # all scales of the level in one launch: the bytes of each query
        b, n, m, radii, nss = head[:5]
        return ("ball_query N=%d M=%d r=%s ns=%s" % (n, m, "+".join("%g" % r for r in radii), "+".join(str(x) for x in nss)),
                sum(b * (n * 12 + m * 12 + m * ns * 4) for ns in nss))
    if name.startswith("ball_query"):
        b, n, m, _r, ns = head[:5]
        return "ball_query N=%d M=%d r=%g ns=%d" % (n, m, _r, ns), b * (n * 12 + m * 12 + m * ns * 4)
    if name.startswith("group_points_w"):
        b, c, n, m, ns = head[:5]
        return ("group_xyz" if c == 3 else "group_feat"), b * (m * ns * 4 + c * n * 4 + c * m * ns * 4)
    if name.startswith("group_concat_multi"):  # both scales of a level in one call: the bytes of each grouping
        b, c, n, m, nss = head[:5]
        total = 0
        for ns in nss:
            total += b * (m * ns * 4 + 3 * n * 4 + 3 * m * ns * 4) + (b * (m * ns * 4 + c * n * 4 + c * m * ns * 4) if c else 0)
        return "group", total
    if name.startswith("group_concat_w"):  # the two grouping calls + centre subtraction + concat of the reference, one output
        b, c, n, m, ns = head[:5]
        xyz_part = b * (m * ns * 4 + 3 * n * 4 + 3 * m * ns * 4)
        return "group", xyz_part + (b * (m * ns * 4 + c * n * 4 + c * m * ns * 4) if c else 0)
    if name.startswith("three_nn"):
        b, n, m = head[:3]
        return "three_nn", b * (n * 12 + m * 12 + n * 24)
    if name.startswith("three_interpolate_w"):
        b, c, m, n = head[:4]
        return "three_interpolate", b * (c * m * 4 + n * 24 + c * n * 4)
    return name, 0


# kernels behind each op family (rocprofv3 names), for the PMC lookup and for the reader of the JSON line
FAMILY_KERNELS = {
    "fps N=16384 M=4096": ["epnet::pruned::fps_indexed_kernel<8, 32, false>"],
    "fps N=4096 M=1024": ["epnet::pruned::fps_indexed_kernel<4, 16, false>"],
    "group": ["epnet::group_xyz_centred_vec4_kernel", "epnet::gather_rows_lds2_kernel<true>", "epnet::gather_rows_lds_kernel<true>"],
    "scene_index N=16384": ["epnet::bq_index_kernel<1024, 14>"],
}
def _newest_pmc_profile():
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic.json")))
    return os.path.relpath(found[-1], ROOT) if found else os.path.join("profiles", "r01_pmc_traffic.json")


PMC_PROFILE = _newest_pmc_profile()   # the committed PMC passes of the latest round


def pmc_traffic(family, args):
    """HBM bytes per launch of an op family from the committed PMC profile (two --pmc passes of this bench with every
    kernel alone on the device, profiles/collect.sh + profiles/summarize.py), scaled by scenes per launch.
    None when no matching profile exists."""
    path = os.path.join(ROOT, PMC_PROFILE)
    if args.kind != "kitti" or args.points != 16384 or not os.path.exists(path):
        return None
    prof = json.load(open(path))
    k = prof.get("families", {}).get(family)
    return None if k is None else int(k["hbm_bytes_avg"] / prof["scenes_per_launch"] * args.batch)


def launch_check(args):
    """--launch-check: what the N-rank launch built, without touching the GPU (tests/test_bench_launch.py runs it on gloo)"""
    import torch.distributed as dist
    from epnet_amd import scene_shard
    rank, local, world = scene_shard.env_world()
    backend, device = os.environ.get("EPNET_BENCH_BACKEND", "nccl"), "cpu"
    if world > 1:
        if backend == "nccl":   # RCCL counts the ranks: one device per rank
            import torch
            device = torch.device("cuda", int(os.environ.get("EPNET_BENCH_DEVICE", local)))
            torch.cuda.set_device(device)
        scene_shard.init_process_group(backend, device=None if device == "cpu" else device)
    seen = int(scene_shard.sum_over_ranks(1.0, device=device))
    idents = scene_shard.gather_over_ranks(scene_shard.device_identity(device))
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "ranks_seen": seen, "gpus_flag": args.gpus,
                          "devices_distinct": scene_shard.distinct_devices(idents), "devices": idents}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    args = parse()
    # --gpus N without a torch.distributed.run environment: start the N ranks as child processes BEFORE this process
    # touches the GPU (it never does), and leave with their exit code
    from epnet_amd import scene_shard
    code = scene_shard.launch_or_continue(args.gpus, os.path.abspath(__file__), sys.argv[1:])
    if code is not None:
        sys.exit(code)
    scene_shard.assert_world(args.gpus)
    if args.launch_check:
        return launch_check(args)
    if not args.rehearsal and ("EPNET_BENCH_DEVICE" in os.environ or "EPNET_BENCH_BACKEND" in os.environ):
        raise SystemExit("EPNET_BENCH_DEVICE / EPNET_BENCH_BACKEND rehearse the N-rank path on one device: pass --rehearsal "
                         "(a measurement line must come from one GPU per rank over RCCL)")
    import torch
    import torch.distributed as dist

    rank, local_rank, world = scene_shard.env_world()
    # CPU baselines first: the multi-core one forks workers, which must happen before this process touches the GPU
    cpu = cpu_multi = None
    if rank == 0 and world == 1 and args.cpu_scenes > 0:
        cpu_multi = cpu_baseline_multicore(args.kind, args.points, max(1, min(16, os.cpu_count() or 1)), args.cfg,
                                           per_proc=2 if args.config == 2 else 1)
        cpu = cpu_baseline(args.kind, args.points, args.cpu_scenes, args.cfg)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the hot path has no CPU fallback)")
    # one process per GPU. EPNET_BENCH_DEVICE / EPNET_BENCH_BACKEND exist only to rehearse the N > 1 code
    # path on a one-GPU box (all ranks on one device, gloo instead of RCCL)
    dev_index = int(os.environ.get("EPNET_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        scene_shard.init_process_group(os.environ.get("EPNET_BENCH_BACKEND", "nccl"), device=dev)

    from epnet_amd import _lib, pointnet2_cuda as ext, sa_stack, synth
    _lib.lib()  # fail loudly if the HIP library is missing

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            scene_shard.barrier()
            torch.cuda.synchronize()

    def time_stack(batch, steps, warmup, cfg=None, with_fp=None, pipelined=None, kind=None):
        """returns (seconds for `steps` steps, the stack, its input batches, the step function). Two DIFFERENT resident batches
        rotate through the steps (batch A on even steps, B on odd ones): a software-pipelined step samples one of them beside
        the grouping of the other, as it does in a loop that consumes a new batch per iteration (tools/train_rcnn.py:221-223)."""
        cfg = args.cfg if cfg is None else cfg
        with_fp = args.with_fp if with_fp is None else with_fp
        pipelined = bool(args.pipelined) if pipelined is None else pipelined
        kind = args.kind if kind is None else kind
        points = args.points if cfg is args.cfg else cfg["n"]
        ids = scene_shard.scene_ids(batch * world, rank, world)           # round-robin shard of the global batch
        stages = args.stages if pipelined else 1
        batches = [torch.stack([synth.cloud(kind, points, scene_shard.scene_seed(1 + which, i)) for i in ids]).to(dev)
                   for which in range(3 if stages == 3 else 2)]            # inputs resident in HBM
        stack = sa_stack.SAStack(batch, n=points, device=dev, with_fp=with_fp, seed=rank,
                                 npoints=cfg["npoints"], radii=cfg["radii"], nsamples=cfg["nsamples"],
                                 feat_channels=cfg["feat_channels"],
                                 overlap=not args.no_overlap, fused=not args.unfused,
                                 shared_index=not args.no_shared_index, pipelined=pipelined,
                                 fused_sampling=not args.module_sampling, stages=stages)
        count = [0]
        if args.no_graph:
            def step():
                stack.step(batches[count[0] % len(batches)])
                count[0] += 1
        elif pipelined:
            stack.capture(*batches[:stack.ring])    # the resident input buffers ARE the batches: no copy per step
            step = stack.replay
        else:
            stack.capture(batches[0])

            def step():                              # every step alone: the one input buffer takes the next batch
                stack.replay(batches[count[0] % len(batches)])
                count[0] += 1
        for _ in range(warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        barrier()
        return time.perf_counter() - t0, stack, batches, step

    def verify_stack(stack, batches, step, scenes):
        """the buffers the timed steps left behind against the oracle, then ONE more step and the same again: the second pass
        sees the other batch in every role (sampled / grouped), so each of the two distinct inputs is verified end to end"""
        mism = {}
        nb = len(batches)
        for which in range(stack.ring if stack.pipelined else 2):
            torch.cuda.synchronize()
            prev2 = None
            if stack.pipelined and not args.no_graph:
                if stack.stages == 3:
                    s1, s2, g = stack.owners3()
                    cur, prev, prev2 = stack.inputs[s1], stack.inputs[s2], stack.inputs[g]
                else:
                    s_par, g_par = stack.owners()
                    cur, prev = stack.inputs[s_par], stack.inputs[g_par]
            else:   # the last step consumed batches[(steps - 1) % nb]; pipelined eager steps worked on the ones before it too
                done = stack.replays if stack.pipelined else step_count(step)
                cur = batches[(done - 1) % nb]
                prev = batches[(done - 2) % nb] if stack.pipelined else None
                prev2 = batches[(done - 3) % nb] if stack.stages == 3 else None
            for s_ in scenes:
                bad = verify_scene(stack, cur, s_, prev_xyz=prev, prev2_xyz=prev2)
                if bad:
                    mism["pass %d scene %d" % (which, s_)] = bad
            step()
        torch.cuda.synchronize()
        return mism

    def step_count(step):
        cells = [c.cell_contents for c in (step.__closure__ or ()) if isinstance(c.cell_contents, list) and len(c.cell_contents) == 1]
        return cells[0][0] if cells else 0

    def dispersion(step, steps):
        """per-step durations from an event pair around every step (recorded in a loop of its own, after the timed one)"""
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        evs[0].record()
        for k in range(steps):
            step()
            evs[k + 1].record()
        torch.cuda.synchronize()
        ms = sorted(evs[k].elapsed_time(evs[k + 1]) for k in range(steps))
        return {"ms_per_step_min": round(ms[0], 4), "ms_per_step_median": round(ms[len(ms) // 2], 4), "ms_per_step_max": round(ms[-1], 4)}

    elapsed, stack, batches, step = time_stack(args.batch, args.steps, args.warmup)
    xyz = batches[0]
    stack_s_levels, stack_chain, stack_stages = stack.s_query_levels, stack.chain, stack.stages
    reduce_dev = dev if dist.is_initialized() and dist.get_backend() == "nccl" else "cpu"
    elapsed_own = elapsed
    elapsed = scene_shard.max_over_ranks(elapsed, device=reduce_dev)
    points_per_step = world * args.batch * args.points
    value = points_per_step * args.steps / elapsed
    ranks_seen = int(scene_shard.sum_over_ranks(1.0, device=reduce_dev))   # the ranks the collective library saw
    my_ms = elapsed_own / args.steps * 1e3
    per_rank = scene_shard.gather_over_ranks({"ms_per_step": round(my_ms, 4), "device": scene_shard.device_identity(dev)})
    devices_distinct = scene_shard.distinct_devices([r_["device"] for r_ in per_rank])
    spread = dispersion(step, args.steps)
    identity_share = stack.chain_identity_share()

    # ---- what the timed steps left in the buffers, against the oracle (rank 0, outside every timed region)
    verification = None
    if rank == 0 and args.verify_scenes > 0:
        picks = sorted({0, args.batch - 1} if args.verify_scenes > 1 else {0})
        picks += [s_ for s_ in range(1, args.batch - 1)][:max(0, args.verify_scenes - len(picks))]
        mism = verify_stack(stack, batches, step, picks)
        verification = {"verified": not mism, "scenes": picks, "distinct_inputs": len(batches), "mismatches": mism,
                        "checked": "fps_idx, centres, ball-query idx and grouped tensors of all %d levels%s of the timed buffers "
                                   "(%s; different resident batches rotate through the steps, each checked in every role -- sampled, "
                                   "queried, grouped) vs oracle/epnet_oracle.c: identical"
                                   % (len(stack.levels), " + three_nn / three_interpolate (1e-5)" if args.with_fp else "",
                                      "pipelined HIP-graph replays" if (args.pipelined and not args.no_graph) else
                                      ("HIP-graph replays" if not args.no_graph else "eager steps"))}

    # ---- per-kernel durations: the same K steps replayed eagerly with HIP events around every launch
    stack.overlap = False  # single stream, unpipelined here, so that an event pair brackets exactly its own kernel
    with OpTimer(torch, ext) as timer:
        for _ in range(args.steps):
            stack.run(xyz)
        torch.cuda.synchronize()
    fam = {}
    for name, head, e0, e1 in timer.records:
        label, nbytes = op_family(name, head)
        f = fam.setdefault(label, {"ms": 0.0, "bytes": 0, "launches": 0, "shapes": {}})
        ms = e0.elapsed_time(e1)
        f["ms"] += ms
        f["bytes"] += nbytes
        f["launches"] += 1
        if label in ("three_nn", "three_interpolate", "group"):   # several shapes under one label: keep them apart as well
            shape = " ".join(str(v) for v in head[:5] if isinstance(v, (int, float, list, tuple)))
            g = f["shapes"].setdefault(shape, {"ms": 0.0, "bytes": 0, "launches": 0})
            g["ms"] += ms
            g["bytes"] += nbytes
            g["launches"] += 1
    kernels = {}
    for label, f in fam.items():
        avg_ms = f["ms"] / f["launches"]
        gbs = (f["bytes"] / f["launches"]) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        kernels[label] = {"launches_per_step": f["launches"] // args.steps, "avg_ms": round(avg_ms, 5),
                          "step_ms": round(f["ms"] / args.steps, 5),
                          "bytes_per_launch": f["bytes"] // f["launches"], "GBps": round(gbs, 2)}
        if f["shapes"]:
            kernels[label]["by_shape"] = {
                shape: {"avg_ms": round(g["ms"] / g["launches"], 5),
                        "GBps": round(g["bytes"] / g["launches"] / (g["ms"] / g["launches"] * 1e-3) / 1e9, 2) if g["ms"] > 0 else 0.0}
                for shape, g in f["shapes"].items()}
    def roof(label, note):
        k = kernels[label]
        traffic = pmc_traffic(label, args)
        return {"bound": "hbm", "kernel": label, "kernel_symbols": FAMILY_KERNELS.get(label), "achieved": k["GBps"],
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(k["GBps"] / HBM_PEAK_GBS, 6), "traffic": traffic,
                "traffic_source": (PMC_PROFILE + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, scaled by scenes per launch)")
                if traffic else None, "avg_launch_ms": k["avg_ms"], "note": note}

    longest, dominant = pick_dominant(kernels)
    fps_note = ("FPS is a chain of M-1 dependent arg-max rounds per scene (latency bound, one workgroup per scene, ~0.6 us per "
                "round): its HBM fraction is tiny by construction; see roofline_hbm_bound for the bandwidth-bound kernel")
    grp_note = ("grouping = [grouped xyz - centre ; grouped features] of BOTH MSG scales of a level (epnet_group_concat_multi: one "
                "call per level, 4 per step, the level-1 one without features; per call the centred-xyz kernel of each scale and "
                "one row gather serving both scales): feature rows staged in LDS once, random reads from LDS, 16-byte coalesced "
                "streaming writes; a plain device-to-device copy on this box runs at device_copy_GBps")
    roofline = roof(dominant, fps_note if dominant.startswith("fps") else grp_note)
    roofline["dominant_by_time"] = {"kernel": longest, "step_ms": kernels[longest]["step_ms"],
                                    "chosen_step_ms": kernels[dominant]["step_ms"]}
    hbm_label = max((k for k in kernels if not k.startswith("fps")), key=lambda k: kernels[k]["step_ms"])
    roofline_hbm = roof(hbm_label, grp_note if hbm_label.startswith("group") else "")
    stack_bytes = (sa_stack.sa_algorithmic_bytes(args.points, args.cfg["npoints"], args.cfg["nsamples"], args.cfg["feat_channels"])["total"]
                   + (sa_stack.fp_algorithmic_bytes()["total"] if args.with_fp else 0))
    stack_gbs = (value / args.points) * stack_bytes / 1e9 / world

    sweep = {}
    for bsz in [int(x) for x in args.sweep.split(",") if x]:
        e = time_stack(bsz, args.steps, args.warmup)[0]
        sweep[str(bsz)] = {"ms_per_step": round(e / args.steps * 1e3, 4), "points_per_s": round(bsz * args.points * args.steps / e, 1)}

    # ---- further lines of the same run (N = 1, default configuration only), each verified against the oracle like the
    # headline: the single-scene latency, the step with the FP ops, and BASELINE config 5
    extras = {}
    if world == 1 and args.config == 2 and not args.with_fp and args.extras:
        del stack, xyz, batches, step
        torch.cuda.empty_cache()

        def extra(name, batch, steps, cfg=None, with_fp=False, pipelined=None, note="", kind=None, into=None):
            cfg_ = args.cfg if cfg is None else cfg
            e, st, bt, stp = time_stack(batch, steps, max(2, args.warmup // 2), cfg=cfg_, with_fp=with_fp, pipelined=pipelined, kind=kind)
            torch.cuda.synchronize()
            share = st.chain_identity_share()
            bad = verify_stack(st, bt, stp, [0]) if args.verify_scenes > 0 else None
            pts = cfg_["n"]
            nbytes = (sa_stack.sa_algorithmic_bytes(pts, cfg_["npoints"], cfg_["nsamples"], cfg_["feat_channels"])["total"]
                      + (sa_stack.fp_algorithmic_bytes()["total"] if with_fp else 0))
            rate = batch * pts * steps / e
            (extras if into is None else into)[name] = {
                "scenes_per_gpu": batch, "points_per_scene": pts, "steps": steps, "ms_per_step": round(e / steps * 1e3, 4),
                "points_per_s": round(rate, 1), "stack_algorithmic_GBps": round(rate / pts * nbytes / 1e9, 2),
                "stack_hbm_frac": round(rate / pts * nbytes / 1e9 / HBM_PEAK_GBS, 6),
                "verified": None if bad is None else not bad, "mismatches": bad or {}, "distinct_inputs": len(bt),
                "identity_share_levels_2_up": share, "workload": note}
            del st, bt, stp
            torch.cuda.empty_cache()

        names = [x for x in args.extras.split(",") if x]
        if "b1" in names:
            extra("latency_one_scene", 1, args.steps, pipelined=False,
                  note="ONE 16384-point scene through the 4-level stack, every step alone (HIP-graph replay): the latency figure")
        if "with_fp" in names:
            extra("with_fp", args.batch, args.steps, with_fp=True,
                  note="the headline step + the 4 three_nn + 4 three_interpolate of the FP modules (SA+FP = 88 087 040 B per scene)")
        if "kinds" in names:   # the other input families of BASELINE.md section 3 config 2, the same pipelined 256-scene step
            extras["kinds"] = {}
            for kind, what in (("ubox", "uniform box (PC_AREA_SCOPE)"),
                               ("dup", "12000 kitti-like points padded to 16384 by re-drawing rows (kitti_rcnn_dataset.py:338-342): exact twins"),
                               ("kitti_q", "kitti-like coordinates rounded to 1e-3 m (velodyne resolution)")):
                extra(kind, args.batch, args.steps, kind=kind, note=what, into=extras["kinds"])
        if "config5" in names:
            extra("config5", 256, max(3, args.steps // 4), cfg=sa_stack.CONFIGS[5],
                  note="BASELINE config 5: dense 65536-point kitti-like scenes, one level -- scene index, FPS 16384, ball query "
                       "r 0.5 / nsample 64, grouping of coordinates and 64 feature channels (python bench.py --config 5 for the full line)")

    # achievable secondary denominator: a plain device-to-device copy of 2 GiB on this box
    src = torch.empty((1 << 29,), dtype=torch.float32, device=dev)
    dst = torch.empty_like(src)
    dst.copy_(src)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize()
    copy_gbs = 5 * 2 * src.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9   # bytes read + bytes written
    del src, dst

    if rank == 0:
        levels = len(args.cfg["npoints"])
        if args.config == 2:
            shape = ("the 4-level SA op stack (4 samplings [FPS + centre gather; levels 2-4 nested in level 1 where tie-free] + 8 ball_query "
                     "+ 8 fused groupings [xyz - centre ; features]%s), "
                     "pyramid 16384>4096>1024>256>64, radii [[.1,.5],[.5,1],[1,2],[2,4]], nsample [16,32], C=0/96/256/512"
                     % (" + 4 three_nn + 4 three_interpolate" if args.with_fp else ""))
        else:
            shape = ("one SA level (BASELINE config 5): scene index, FPS %d>%d, gather, ball_query r=%g nsample=%d, fused grouping "
                     "[xyz - centre ; %d feature channels]" % (args.points, args.cfg["npoints"][0], args.cfg["radii"][0][0],
                                                               args.cfg["nsamples"][0][0], args.cfg["feat_channels"][0]))
        line = {
            "metric": ("SA-stack points/sec per GPU (16384-pt KITTI scene) + % HBM roofline" if args.config == 2 else
                       "SA-level points/sec per GPU (dense 65536-pt scene, BASELINE config 5) + % HBM roofline"),
            "value": round(value, 1), "unit": "points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%d x %d-pt %s scenes per GPU per step through %s, %s launch"
                                   % (args.batch, args.points, args.kind, shape, "eager" if args.no_graph else "HIP-graph"),
                       "baseline_config": args.config, "levels": levels,
                       "scenes_per_gpu": args.batch, "software_pipelined": bool(args.pipelined), "pipeline_stages": stack_stages,
                       "ball_queries_in_stage_s_levels": [v + 1 for v in sorted(stack_s_levels)], "points_per_scene": args.points, "parallelism": "scene-parallel x%d" % world},
            "points_per_s_per_gpu": round(value / world, 1),
            "stack_algorithmic_GBps_per_gpu": round(stack_gbs, 2), "stack_hbm_frac": round(stack_gbs / HBM_PEAK_GBS, 6),
            "device_copy_GBps": round(copy_gbs, 1), "stack_frac_of_device_copy": round(stack_gbs / copy_gbs, 6),
            "roofline": roofline, "roofline_hbm_bound": roofline_hbm, "kernels": kernels, "cpu_baseline": cpu,
            "cpu_baseline_multicore": cpu_multi,
            "verified": None if verification is None else verification["verified"], "verification": verification,
            "ranks_seen": ranks_seen, "devices_distinct": devices_distinct, "rehearsal": bool(args.rehearsal),
            "per_rank_ms_per_step": {"min": min(r_["ms_per_step"] for r_ in per_rank), "max": max(r_["ms_per_step"] for r_ in per_rank),
                                     "ranks": [r_["ms_per_step"] for r_ in per_rank]},
            "devices": [r_["device"] for r_ in per_rank],
            "sampling_chain": {"enabled": bool(stack_chain), "identity_share_levels_2_up": identity_share,
                               "note": "levels 2.. sample the centres of the level above: a scene whose tie-free prefix covers the "
                                       "level's sample count takes idx = 0..m-1 without running rounds (epnet_sample_centres_chain)"},
        }
        line.update(spread)
        if world > 1:
            line["cpu_baseline_note"] = "the CPU baseline is timed in the N = 1 run only (rank 0 there has the host to itself)"
        if sweep:
            line["sweep"] = sweep
        line.update(extras)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
