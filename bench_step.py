#!/usr/bin/env python3
"""bench_step.py -- BASELINE.json config 4, the per-rank part: one `rcnn_online` training step of the POINT stream at
the yaml's shapes (tools/cfgs/LI_Fusion_with_attention_use_ce_loss.yaml) on synthetic KITTI-shaped scenes.

    python bench_step.py [--batch 2] [--steps 10] [--warmup 3] [--points 16384]
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 bench_step.py     # scene-parallel, RCCL

What is timed (forward + backward + SGD step, `--batch` scenes per GPU):
  RPN backbone   4 SA-MSG levels 16384>4096>1024>256>64 + 4 FP levels (lib/net/pointnet2_msg.py:126-196, point stream) and
                 the cls / reg heads (lib/net/rpn.py:23-52)
  proposals      ProposalLayer (decode, distance-based NMS, 512 proposals per scene)
  targets        ProposalTargetLayer (IoU, ROI sampling + augmentation, roipool3d 64 x 512 x 133, canonical transform)
  RCNN stage     3 SA levels over 64 ROIs x 512 points per scene (128 > 32 > group-all) and two FC heads
                 (lib/net/rcnn_net.py:43-93, without its loss bookkeeping)
Every geometry op is this package's HIP path; every dense layer (shared MLPs, batch norm, heads, optimiser) is stock
PyTorch-ROCm, as the north_star has it. The image stream / LI-Fusion of configs 3-4 is the reference's stock-PyTorch
territory and is not part of this harness; losses are placeholders (sums of squares) -- the step exercises autograd
through every op of the hot path at the real shapes, it does not train anything. With N > 1 ranks the model is wrapped
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


def build_model(scale=1, rpn_channels=76):
    """the two-stage point model; scale > 1 divides the pyramid's point counts (small test configurations)"""
    import torch
    import torch.nn as nn
    from epnet_amd import pointnet2_utils as p2u, pytorch_utils as pt_utils
    from epnet_amd.pointnet2_modules import PointnetFPModule, PointnetSAModule, PointnetSAModuleMSG
    PYRAMID = os.environ.get("EPNET_SA_PYRAMID", "1") != "0"

    class Backbone(nn.Module):   # lib/net/pointnet2_msg.py:126-196, 201-232 without the image branch
        def __init__(self):
            super().__init__()
            self.SA_modules, self.FP_modules = nn.ModuleList(), nn.ModuleList()
            channel_in, skips = 0, [0]
            for k in range(len(SA_NPOINTS)):
                mlps = [[channel_in] + list(m) for m in SA_MLPS[k]]
                self.SA_modules.append(PointnetSAModuleMSG(npoint=SA_NPOINTS[k] // scale, radii=SA_RADIUS[k], nsamples=SA_NSAMPLE[k],
                                                           mlps=mlps, use_xyz=True, bn=True))
                channel_in = sum(m[-1] for m in mlps)
                skips.append(channel_in)
            for k in range(len(FP_MLPS)):
                pre = FP_MLPS[k + 1][-1] if k + 1 < len(FP_MLPS) else channel_in
                self.FP_modules.append(PointnetFPModule(mlp=[pre + skips[k]] + FP_MLPS[k]))

        def forward(self, xyz):
            l_xyz, l_feat = [xyz], [None]
            # every level's sampling up front on a side stream: it runs beside the MLPs instead of between them
            pyramid = p2u.sample_pyramid(xyz, [sa.npoint for sa in self.SA_modules]) if PYRAMID else [None] * len(self.SA_modules)
            for sa, pre in zip(self.SA_modules, pyramid):
                nx, nf, _ = sa(l_xyz[-1], l_feat[-1], presampled=pre)
                l_xyz.append(nx)
                l_feat.append(nf)
            for i in range(-1, -(len(self.FP_modules) + 1), -1):
                l_feat[i - 1] = self.FP_modules[i](l_xyz[i - 1], l_xyz[i], l_feat[i - 1], l_feat[i])
            return l_feat[0]

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
            self.backbone = Backbone()
            self.rpn_cls = nn.Sequential(pt_utils.Conv1d(128, 128, bn=True), pt_utils.Conv1d(128, 1, activation=None))
            self.rpn_reg = nn.Sequential(pt_utils.Conv1d(128, 128, bn=True), pt_utils.Conv1d(128, rpn_channels, activation=None))
            self.rcnn = RCNN()
            self.layers = None   # (ProposalLayer, ProposalTargetLayer), set by the caller

        def forward(self, xyz, gt_boxes3d, mark=None):
            """one forward pass + placeholder loss (the whole step lives in forward so that DistributedDataParallel sees
            it); returns (loss, dict of outputs)"""
            proposal_layer, target_layer = self.layers
            mark = mark if mark is not None else (lambda name: None)
            feats = self.backbone(xyz)                                           # (B,128,N)
            rpn_cls = self.rpn_cls(feats).transpose(1, 2).contiguous()           # (B,N,1)
            rpn_reg = self.rpn_reg(feats).transpose(1, 2).contiguous()           # (B,N,76)
            mark("rpn")
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


def run_step(model, layers, xyz, gt_boxes3d, timer=None):
    """one forward pass through `model` (plain or DistributedDataParallel-wrapped)"""
    (model.module if hasattr(model, "module") else model).layers = layers
    return model(xyz, gt_boxes3d, timer)


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
            feats = model.backbone(xyz)
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
    ap.add_argument("--gpus", type=int, default=1,
                    help="ranks (one per GPU); without a torch.distributed.run environment the ranks are started as child processes")
    ap.add_argument("--launch-check", action="store_true", help="rehearse the N-rank launch only (see bench.py)")
    args = ap.parse_args()
    from epnet_amd import scene_shard
    code = scene_shard.launch_or_continue(args.gpus, os.path.abspath(__file__), sys.argv[1:])
    if code is not None:
        sys.exit(code)
    scene_shard.assert_world(args.gpus)
    if args.launch_check:
        import bench
        return bench.launch_check(args)
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
    model = build_model().to(device)
    if world > 1:
        model = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local], find_unused_parameters=False)
    opt = torch.optim.SGD(model.parameters(), lr=1e-4, momentum=0.9)
    layers = (pl.ProposalLayer("TRAIN").to(device), ptl.ProposalTargetLayer())
    xyz, gts = synthetic_batch(args.batch, args.points, 100 + 1000 * rank, device)   # every rank its own scenes

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
        loss, _ = run_step(model, layers, xyz, gts, mark if timed else None)
        loss.backward()
        mark("backward")
        opt.step()
        mark("optimizer")
        if timed:
            torch.cuda.synchronize()
            for (_, a), (name, b) in zip(events[:-1], events[1:]):
                phases[name] = phases.get(name, 0.0) + a.elapsed_time(b)
        return float(loss.detach()) if timed else None

    for _ in range(args.warmup):
        one(False)
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
    if world > 1:
        t = torch.tensor([elapsed], device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    if rank == 0:
        n_param = sum(p.numel() for p in model.parameters())
        print(json.dumps({"metric": "rcnn_online point-stream training step (BASELINE config 4, per-rank part)", "n_gpus": world,
                          "scenes_per_gpu": args.batch, "points_per_scene": args.points, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(elapsed / args.steps * 1e3, 3),
                          "ms_per_step_median": round(sorted(per_step)[len(per_step) // 2] * 1e3, 3),
                          "ms_per_step_max": round(max(per_step) * 1e3, 3),
                          "scenes_per_s": round(world * args.batch * args.steps / elapsed, 2),
                          "phase_ms": {k: round(v / args.steps, 3) for k, v in phases.items()},
                          "parameters": n_param, "grad_allreduce_MB": round(n_param * 4 / 1e6, 1) if world > 1 else 0,
                          "loss": last, "data": "synthetic", "dtype": "f32",
                          "note": "phase_ms from HIP events (phases are host-serialised by the two syncs of the target layer); "
                                  "dense layers are stock PyTorch-ROCm, geometry ops this package's HIP kernels"}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
