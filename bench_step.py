#!/usr/bin/env python3
"""bench_step.py -- BASELINE.json configs 3 and 4: training steps of the model AROUND the hot path at the yaml's shapes
(tools/cfgs/LI_Fusion_with_attention_use_ce_loss.yaml) on synthetic KITTI-shaped scenes.

    python bench_step.py --image --rpn-only [--batch 2]       # config 3: two-stream RPN (point + image stream, LI-Fusion) fwd+bwd
    python bench_step.py --image [--gpus N]                   # config 4: the whole rcnn_online step, 62.7 MB of gradients
    python bench_step.py                                      # the point stream alone (round-1 figure)
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 bench_step.py --gpus N   # scene-parallel, RCCL

What is timed (forward + backward + SGD step, `--batch` scenes per GPU):
  RPN backbone   4 SA-MSG levels 16384>4096>1024>256>64 + 4 FP levels (lib/net/pointnet2_msg.py:126-196, point stream) and
                 the cls / reg heads (lib/net/rpn.py:23-52)
  proposals      ProposalLayer (decode, distance-based NMS, 512 proposals per scene)
  targets        ProposalTargetLayer (IoU, ROI sampling + augmentation, roipool3d 64 x 512 x 133, canonical transform)
  RCNN stage     3 SA levels over 64 ROIs x 512 points per scene (128 > 32 > group-all) and two FC heads
                 (lib/net/rcnn_net.py:43-93, without its loss bookkeeping)
Every geometry op is this package's HIP path; every dense layer (shared MLPs, batch norm, image convolutions, attention
fusion, heads, optimiser) is stock PyTorch-ROCm, as the north_star has it. With --image the backbone is the two-stream one
(epnet_amd/rpn_backbone.py = lib/net/pointnet2_msg.py: four strided conv blocks over a (B,3,384,1280) image ~N(0,1),
Feature_Gather at the four pyramid levels and at full resolution, attention fusion); pixel coordinates ~U[0,1280)xU[0,384).
Losses are placeholders (sums of squares) -- the step exercises autograd through every op of the hot path at the real
shapes, it does not train anything. `ops_share` = HIP-event time inside this package's operators / the step's GPU time,
from an instrumented pass after the timed one. With N > 1 ranks the model is wrapped
in DistributedDataParallel: the gradient all-reduce over RCCL / xGMI is the step's only collective (scenes are sharded
by rank). One JSON line from rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SA_NPOINTS = [4096, 1024, 256, 64]
SA_RADIUS = [[0.1, 0.5], [0.5, 1.0], [1.0, 2.0], [2.0, 4.0]]
SA_NSAMPLE = [[16, 32], [16, 32], [16, 32], [16, 32]]
SA_MLPS = [[[16, 16, 32], [32, 32, 64]], [[64, 64, 128], [64, 96, 128]], [[128, 196, 256], [128, 196, 256]],
           [[256, 256, 512], [256, 384, 512]]]
FP_MLPS = [[128, 128], [256, 256], [512, 512], [512, 512]]
RCNN_NPOINTS, RCNN_RADIUS, RCNN_NSAMPLE = [128, 32, None], [0.2, 0.4, 100], [64, 64, 64]
RCNN_MLPS = [[128, 128, 128], [128, 128, 256], [256, 256, 512]]


def build_model(scale=1, rpn_channels=76, image=False, sampler="hip"):
    """the two-stage model; scale > 1 divides the pyramid's point counts (small test configurations); image: the
    two-stream backbone with LI-Fusion (configs 3 / 4) instead of the point stream alone"""
    import torch
    import torch.nn as nn
    from epnet_amd import pytorch_utils as pt_utils, rpn_backbone
    from epnet_amd.pointnet2_modules import PointnetSAModule
    PYRAMID = os.environ.get("EPNET_SA_PYRAMID", "1") != "0"

    class RCNN(nn.Module):       # lib/net/rcnn_net.py:17-93 (xyz up-layer, merge, SA stack, heads), losses left out
        def __init__(self):
            super().__init__()
            self.xyz_up = pt_utils.SharedMLP([5, 128, 128], bn=False)          # xyz + mask + depth (:22-27)
            self.merge_down = pt_utils.SharedMLP([256, 128], bn=False)
            self.SA_modules = nn.ModuleList()
            cin = 128
            for k in range(len(RCNN_NPOINTS)):
                self.SA_modules.append(PointnetSAModule(npoint=RCNN_NPOINTS[k], radius=RCNN_RADIUS[k], nsample=RCNN_NSAMPLE[k],
                                                        mlp=[cin] + RCNN_MLPS[k], use_xyz=True, bn=False))
                cin = RCNN_MLPS[k][-1]
            self.cls = nn.Sequential(pt_utils.Conv1d(512, 512), pt_utils.Conv1d(512, 512), pt_utils.Conv1d(512, 1, activation=None))
            self.reg = nn.Sequential(pt_utils.Conv1d(512, 512), pt_utils.Conv1d(512, 512), pt_utils.Conv1d(512, 46, activation=None))

        def forward(self, pts, feats):
            """pts (R,512,3) canonical, feats (R,512,130) = [mask, depth, 128 rpn features] (:96-113)"""
            head = torch.cat((pts, feats[..., 0:2]), dim=2).transpose(1, 2).unsqueeze(3)
            merged = torch.cat((self.xyz_up(head), feats[..., 2:].transpose(1, 2).unsqueeze(3)), dim=1)
            f = self.merge_down(merged).squeeze(3)
            xyz = pts.contiguous()
            for sa in self.SA_modules:
                xyz, f, _ = sa(xyz, f.contiguous())
            return self.cls(f), self.reg(f)

    class TwoStage(nn.Module):
        def __init__(self):
            super().__init__()
            # lib/net/pointnet2_msg.py:126-259 (rpn.py:19: input_channels = 0, the yaml has USE_INTENSITY False)
            self.backbone = rpn_backbone.Pointnet2MSG(input_channels=0, config=rpn_backbone.BackboneConfig(li_fusion=image),
                                                      sampler=sampler, scale=scale, pyramid=PYRAMID)
            self.two_stream = image
            self.rpn_cls = nn.Sequential(pt_utils.Conv1d(128, 128, bn=True), pt_utils.Conv1d(128, 1, activation=None))
            self.rpn_reg = nn.Sequential(pt_utils.Conv1d(128, 128, bn=True), pt_utils.Conv1d(128, rpn_channels, activation=None))
            self.rcnn = RCNN()
            self.layers = None   # (ProposalLayer, ProposalTargetLayer), set by the caller

        def forward(self, xyz, gt_boxes3d, mark=None, image=None, xy=None, rpn_only=False):
            """one forward pass + placeholder loss (the whole step lives in forward so that DistributedDataParallel sees
            it); returns (loss, dict of outputs). image (B,3,H,W) / xy (B,N,2) pixel coordinates (normalised in place, as
            the reference does) for the two-stream backbone"""
            proposal_layer, target_layer = self.layers
            mark = mark if mark is not None else (lambda name: None)
            _, feats = self.backbone(xyz, image, xy) if self.two_stream else self.backbone(xyz)   # (B,128,N)
            rpn_cls = self.rpn_cls(feats).transpose(1, 2).contiguous()           # (B,N,1)
            rpn_reg = self.rpn_reg(feats).transpose(1, 2).contiguous()           # (B,N,76)
            mark("rpn")
            if rpn_only:                                                         # config 3: backbone + heads, dummy loss
                return rpn_cls.pow(2).mean() + rpn_reg.pow(2).mean(), {"rpn_cls": rpn_cls, "rpn_reg": rpn_reg}
            with torch.no_grad():                                                # lib/net/point_rcnn.py:33-47
                scores = rpn_cls[:, :, 0].detach()
                rois, _ = proposal_layer(scores, rpn_reg.detach(), xyz)
                mark("proposals")
                seg_mask = (torch.sigmoid(scores) > 0.3).float()
                depth = torch.norm(xyz, p=2, dim=2)
                target = target_layer({"roi_boxes3d": rois, "gt_boxes3d": gt_boxes3d, "rpn_xyz": xyz,
                                       "rpn_features": feats.detach().permute(0, 2, 1).contiguous(), "seg_mask": seg_mask,
                                       "pts_depth": depth})
                mark("targets")
            rcnn_cls, rcnn_reg = self.rcnn(target["sampled_pts"], target["pts_feature"])
            mark("rcnn")
            loss = rpn_cls.pow(2).mean() + rpn_reg.pow(2).mean() + rcnn_cls.pow(2).mean() + rcnn_reg.pow(2).mean()
            return loss, {"rois": rois, "target": target, "rcnn_cls": rcnn_cls, "rcnn_reg": rcnn_reg}

    return TwoStage()


def run_step(model, layers, xyz, gt_boxes3d, timer=None, image=None, xy=None, rpn_only=False):
    """one forward pass through `model` (plain or DistributedDataParallel-wrapped)"""
    (model.module if hasattr(model, "module") else model).layers = layers
    return model(xyz, gt_boxes3d, timer, image, None if xy is None else xy.clone(), rpn_only)


def synthetic_batch(batch, points, seed, device):
    import torch
    from epnet_amd import synth
    xyz = synth.scenes("kitti", batch, points, seed=seed).to(device)
    gts = torch.zeros((batch, 20, 7))
    for i in range(batch):
        gts[i, :12] = synth.object_boxes(12, seed + i)
    return xyz, gts.to(device)


def infer(args, model, proposal_layer, xyz):
    """RPN-stage inference latency: eager launches against one HIP-graph replay"""
    import torch
    model.eval()

    def stage():
        with torch.no_grad():
            _, feats = model.backbone(xyz)
            cls = model.rpn_cls(feats).transpose(1, 2).contiguous()
            reg = model.rpn_reg(feats).transpose(1, 2).contiguous()
            return proposal_layer(cls[:, :, 0].contiguous(), reg, xyz)

    def timed(fn, reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(max(3, args.warmup)):
            eager_out = stage()
    torch.cuda.current_stream().wait_stream(side)
    ms_eager = timed(stage, args.steps)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        graph_out = stage()
    graph.replay()
    torch.cuda.synchronize()
    same = all(torch.equal(a, b) for a, b in zip(eager_out, graph_out))
    ms_graph = timed(graph.replay, args.steps)
    print(json.dumps({"metric": "RPN-stage inference latency (backbone + heads + proposal layer, eval mode)", "scenes": args.batch,
                      "points_per_scene": args.points, "ms_eager": round(ms_eager, 3), "ms_hip_graph": round(ms_graph, 3),
                      "graph_equals_eager": bool(same), "dtype": "f32", "data": "synthetic"}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2, help="scenes per GPU (16 over 8 GPUs in config 4)")
    ap.add_argument("--points", type=int, default=16384)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--infer", action="store_true",
                    help="instead of the training step: inference latency of the RPN stage (backbone, heads, proposal layer) in "
                         "eval mode, eagerly and replayed from a HIP graph (nothing in the stage synchronises with the host)")
    ap.add_argument("--image", action="store_true",
                    help="two-stream backbone: (B,3,384,1280) image ~N(0,1), pixel coordinates ~U[0,1280)xU[0,384), LI-Fusion with "
                         "image attention (BASELINE configs 3 and 4; 15.7 M parameters = 62.7 MB of gradients)")
    ap.add_argument("--rpn-only", action="store_true", help="config 3: forward + backward of backbone + RPN heads only")
    ap.add_argument("--sampler", default="hip", choices=["hip", "stock"],
                    help="point-to-pixel sampler: this package's Feature_Gather or stock torch.gather + grid_sample")
    ap.add_argument("--gpus", type=int, default=1,
                    help="ranks (one per GPU); without a torch.distributed.run environment the ranks are started as child processes")
    ap.add_argument("--launch-check", action="store_true", help="rehearse the N-rank launch only (see bench.py)")
    ap.add_argument("--rehearsal", action="store_true",
                    help="allow EPNET_BENCH_DEVICE / EPNET_BENCH_BACKEND (all ranks on one device, gloo): see bench.py")
    args = ap.parse_args()
    if args.infer and args.image:
        ap.error("--infer times the point-stream RPN stage; the two-stream backbone (--image) is timed by the training step only")
    from epnet_amd import scene_shard
    code = scene_shard.launch_or_continue(args.gpus, os.path.abspath(__file__), sys.argv[1:])
    if code is not None:
        sys.exit(code)
    scene_shard.assert_world(args.gpus)
    if args.launch_check:
        import bench
        return bench.launch_check(args)
    if not args.rehearsal and ("EPNET_BENCH_DEVICE" in os.environ or "EPNET_BENCH_BACKEND" in os.environ):
        raise SystemExit("EPNET_BENCH_DEVICE / EPNET_BENCH_BACKEND rehearse the N-rank path on one device: pass --rehearsal")
    import numpy as np
    import torch
    import torch.distributed as dist
    from epnet_amd import proposal_layer as pl, proposal_target_layer as ptl

    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    # EPNET_BENCH_DEVICE / EPNET_BENCH_BACKEND rehearse the N > 1 path on a one-GPU box (all ranks on one device, gloo)
    local = int(os.environ.get("EPNET_BENCH_DEVICE", local))
    backend = os.environ.get("EPNET_BENCH_BACKEND", "nccl")
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    torch.manual_seed(1 + rank)
    np.random.seed(1 + rank)
    model = build_model(image=args.image, sampler=args.sampler).to(device)
    if world > 1:
        model = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local], find_unused_parameters=False)
    opt = torch.optim.SGD(model.parameters(), lr=1e-4, momentum=0.9)
    layers = (pl.ProposalLayer("TRAIN").to(device), ptl.ProposalTargetLayer())
    xyz, gts = synthetic_batch(args.batch, args.points, 100 + 1000 * rank, device)   # every rank its own scenes
    image = xy = None
    if args.image:
        g = torch.Generator().manual_seed(200 + rank)
        image = torch.randn((args.batch, 3, 384, 1280), generator=g).to(device)                      # lib/datasets/kitti_dataset.py:54
        xy = (torch.rand((args.batch, args.points, 2), generator=g) * torch.tensor([1280.0, 384.0])).to(device)

    if args.infer:
        return infer(args, model.module if hasattr(model, "module") else model, layers[0], xyz)

    phases = {}

    def one(timed):
        events = [("start", torch.cuda.Event(enable_timing=True))]
        events[0][1].record()

        def mark(name):
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            events.append((name, e))
        opt.zero_grad(set_to_none=True)
        loss, _ = run_step(model, layers, xyz, gts, mark if timed else None, image, xy, args.rpn_only)
        loss.backward()
        mark("backward")
        opt.step()
        mark("optimizer")
        if timed:
            torch.cuda.synchronize()
            for (_, a), (name, b) in zip(events[:-1], events[1:]):
                phases[name] = phases.get(name, 0.0) + a.elapsed_time(b)
        return float(loss.detach()) if timed else None

    t_w = time.perf_counter()
    for k in range(args.warmup):
        one(False)
        if k == 0 and rank == 0:
            torch.cuda.synchronize()
            print("first step (kernel selection of the dense layers included): %.1f s" % (time.perf_counter() - t_w), file=sys.stderr, flush=True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last, per_step = None, []
    for _ in range(args.steps):
        t_step = time.perf_counter()
        last = one(True)
        per_step.append(time.perf_counter() - t_step)      # one(True) ends with a device synchronisation
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    elapsed_own = elapsed
    if world > 1:
        t = torch.tensor([elapsed], device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    per_rank = scene_shard.gather_over_ranks({"ms_per_step": round(elapsed_own / args.steps * 1e3, 3),
                                              "device": scene_shard.device_identity(device)})
    # the gradient exchange on its own: one all-reduce of a buffer as large as all gradients (DDP sends the same bytes in buckets,
    # overlapped with backward -- this is the un-overlapped cost it hides)
    allreduce = None
    if world > 1:
        n_grad = sum(p.numel() for p in model.parameters() if p.requires_grad)
        flat = torch.zeros((n_grad,), dtype=torch.float32, device=device)
        for _ in range(2):
            dist.all_reduce(flat)
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a0.record()
        for _ in range(5):
            dist.all_reduce(flat)
        a1.record()
        torch.cuda.synchronize()
        ar_ms = a0.elapsed_time(a1) / 5
        allreduce = {"bytes": n_grad * 4, "ms_alone": round(ar_ms, 3), "backend": dist.get_backend(),
                     "bus_GBps": round(2 * (world - 1) / world * n_grad * 4 / (ar_ms * 1e-3) / 1e9, 2)}
        del flat
    # ---- share of the step spent inside this package's operators: an instrumented pass with a HIP event pair around every
    # call into the three extension stand-ins (on the stream the call launches on), against the GPU time of the same steps
    ops_share = None
    import contextlib
    import bench
    from epnet_amd import iou3d_cuda, pointnet2_cuda, roipool3d_cuda
    iou_names = [n for n in ("boxes_overlap_bev_gpu", "boxes_iou_bev_gpu", "boxes_iou3d_fused_gpu", "boxes_iou3d_pairs_gpu",
                             "aug_roi_by_noise_gpu", "rpn_proposals_gpu", "nms_device", "nms_normal_device")]
    timers = ([bench.OpTimer(torch, pointnet2_cuda), bench.OpTimer(torch, iou3d_cuda, iou_names),
               bench.OpTimer(torch, roipool3d_cuda, ["forward"])] if rank == 0 else [])
    reps = max(2, min(5, args.steps))
    with contextlib.ExitStack() as stack:       # (every rank takes the steps: the gradient all-reduce needs all of them)
        for t in timers:
            stack.enter_context(t)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            one(False)
        e1.record()
        torch.cuda.synchronize()
    if rank == 0:
        per_op = {}
        for t in timers:
            for name, _head, a, b_ in t.records:
                per_op[name] = per_op.get(name, 0.0) + a.elapsed_time(b_) / reps
        ops_ms = sum(per_op.values())
        step_gpu_ms = e0.elapsed_time(e1) / reps
        ops_share = {"ops_ms_per_step": round(ops_ms, 3), "instrumented_step_ms": round(step_gpu_ms, 3),
                     "share": round(ops_ms / step_gpu_ms, 4),
                     "per_op_ms": {k: round(v, 3) for k, v in sorted(per_op.items(), key=lambda kv: -kv[1])},
                     "note": "sum of HIP-event durations of every call into the pointnet2 / iou3d / roipool3d stand-ins "
                             "(calls on the side stream overlap the dense layers, so the share is of GPU work, not of wall time)"}
    if rank == 0:
        n_param = sum(p.numel() for p in model.parameters())
        what = ("two-stream RPN fwd+bwd (BASELINE config 3)" if (args.image and args.rpn_only) else
                "rcnn_online training step, two-stream model (BASELINE config 4)" if args.image else
                "RPN fwd+bwd, point stream only" if args.rpn_only else
                "rcnn_online point-stream training step (BASELINE config 4 without the image stream)")
        print(json.dumps({"metric": what, "n_gpus": world, "image_stream": bool(args.image), "rpn_only": bool(args.rpn_only),
                          "sampler": args.sampler, "ops_share": ops_share,
                          "scenes_per_gpu": args.batch, "points_per_scene": args.points, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(elapsed / args.steps * 1e3, 3),
                          "ms_per_step_median": round(sorted(per_step)[len(per_step) // 2] * 1e3, 3),
                          "ms_per_step_max": round(max(per_step) * 1e3, 3),
                          "ms_per_step_all": [round(x * 1e3, 2) for x in per_step],
                          "scenes_per_s": round(world * args.batch * args.steps / elapsed, 2),
                          "phase_ms": {k: round(v / args.steps, 3) for k, v in phases.items()},
                          "parameters": n_param, "grad_allreduce_MB": round(n_param * 4 / 1e6, 1) if world > 1 else 0,
                          "grad_allreduce": allreduce, "devices_distinct": scene_shard.distinct_devices([r_["device"] for r_ in per_rank]),
                          "devices": [r_["device"] for r_ in per_rank], "rehearsal": bool(args.rehearsal),
                          "per_rank_ms_per_step": {"min": min(r_["ms_per_step"] for r_ in per_rank),
                                                   "max": max(r_["ms_per_step"] for r_ in per_rank)},
                          "loss": last, "data": "synthetic", "dtype": "f32",
                          "note": "phase_ms from HIP events (phases are host-serialised by the two syncs of the target layer); "
                                  "dense layers are stock PyTorch-ROCm, geometry ops this package's HIP kernels"}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
