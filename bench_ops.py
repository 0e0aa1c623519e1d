#!/usr/bin/env python3
"""bench_ops.py -- per-operator timings of the hot-path rows that are not part of bench.py's SA stack:
iou3d (rotated overlap / IoU, both NMS flavours), roipool3d, three_nn / three_interpolate and the three
gradient ops, at the shapes of the reference's rcnn_online step. One JSON line per op:
median HIP-event time, algorithmic bytes (SURVEY.md 8d formulas) and the resulting GB/s.

    python bench_ops.py [--reps 20]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    import torch
    from epnet_amd import iou3d_cuda, iou3d_utils, kitti_utils, pointnet2_cuda as p2, roipool3d_cuda, synth

    dev = torch.device("cuda:0")
    f32, i32 = torch.float32, torch.int32

    def timeit(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(args.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        return ts[len(ts) // 2]

    def timeit_once(fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1)

    def report(op, shape, ms, nbytes, note=""):
        print(json.dumps({"op": op, "shape": shape, "ms": round(ms, 4), "algorithmic_bytes": nbytes,
                          "GBps": round(nbytes / (ms * 1e-3) / 1e9, 2), "note": note}), flush=True)

    g = torch.Generator().manual_seed(0)
    # ---- NMS at the proposal-layer sizes (RPN.NMS_TYPE normal, N <= 6300 / 2700, thresh 0.85) and eval rotated NMS
    for n, rot, thr in ((6300, False, 0.85), (2700, False, 0.85), (6300, True, 0.8), (512, True, 0.1), (100, True, 0.1)):
        boxes, scores = synth.proposal_boxes(n, seed=n, num_objects=40, jitter=1.5)
        bev, sc = kitti_utils.boxes3d_to_bev_torch(boxes).to(dev), scores.to(dev)
        fn = iou3d_utils.nms_gpu if rot else iou3d_utils.nms_normal_gpu
        kept = fn(bev, sc, thr)
        ms = timeit(lambda: fn(bev, sc, thr))
        srt = bev[sc.sort(0, descending=True)[1]].contiguous()
        dfn = iou3d_cuda.nms_device if rot else iou3d_cuda.nms_normal_device
        ms_dev = timeit(lambda: dfn(srt, thr))
        report("nms_gpu" if rot else "nms_normal_gpu", {"N": n, "thresh": thr, "kept": int(kept.numel())}, ms,
               n * 20 + n * ((n + 63) // 64) * 8, "surface call incl. sort + 4-byte count sync; mask+sweep kernels alone %.4f ms" % ms_dev)
    # ---- 3-D IoU of 512 ROIs x 20 GT boxes, and the 1x1 calls of aug_roi_by_noise
    a, _ = synth.proposal_boxes(512, seed=1)
    b, _ = synth.proposal_boxes(20, seed=2)
    a, b = a.to(dev), b.to(dev)
    report("boxes_iou3d_gpu", {"Na": 512, "Nb": 20}, timeit(lambda: iou3d_utils.boxes_iou3d_gpu(a, b)), 512 * 20 + 20 * 20 + 512 * 20 * 4, "incl. torch height/volume math")
    a1, b1 = a[:1].contiguous(), b[:1].contiguous()
    report("boxes_iou3d_gpu", {"Na": 1, "Nb": 1}, timeit(lambda: iou3d_utils.boxes_iou3d_gpu(a1, b1)), 44, "launch-latency bound")
    # ---- ROI augmentation: 2 scenes x 64 ROIs x up to 10 tries -- one launch against the reference's
    # host loop (proposal_target_layer.py:220-247: per try a single-pair IoU composed of ~15 launches and a device->host
    # read of the result), restated here on this package's own surface for timing
    from epnet_amd import proposal_target_layer as ptl
    k_rois, t_max = 128, 10
    boxes, _ = synth.proposal_boxes(2 * k_rois, seed=3, num_objects=16, jitter=0.6)
    rois0, gts0 = boxes[:k_rois].to(dev).contiguous(), boxes[k_rois:].to(dev).contiguous()
    src0 = torch.rand((k_rois,), generator=g).to(dev)
    tries = torch.tensor([t_max] * 32 + [1] * 32, dtype=i32).repeat(2).to(dev)

    def batched():
        keep, noise = ptl.draw_aug_tables(k_rois, t_max, "multiple", dev)
        return ptl.aug_roi_by_noise_batched(rois0.clone(), gts0, src0, 0.55, keep, noise, tries)

    def host_loop():
        keep, noise = ptl.draw_aug_tables(k_rois, t_max, "multiple", dev)
        keep_h = keep.cpu().numpy()
        out = rois0.clone()
        n_try = tries.cpu().numpy()
        launches = 0
        for k in range(k_rois):
            temp_iou, cnt, aug = 0.0, 0, out[k]
            while temp_iou < 0.55 and cnt < n_try[k]:
                nz = noise[k, cnt]
                aug = out[k] if keep_h[k, cnt] else torch.cat([out[k, 0:3] + nz[0:3], out[k, 3:6] * nz[3:6], out[k, 6:7] + nz[6:7]])
                temp_iou = float(iou3d_utils.boxes_iou3d_composed(aug.view(1, 7), gts0[k:k + 1])[0, 0])
                cnt += 1
                launches += 1
            out[k] = aug
        return launches
    ms_b = timeit(batched)
    n_pairs = host_loop()
    ms_h = timeit(host_loop) if args.reps <= 5 else sorted(timeit_once(host_loop) for _ in range(3))[1]
    report("aug_roi_by_noise", {"rois": k_rois, "aug_times": t_max, "fg": 64, "bg": 64}, ms_b, k_rois * (28 * 2 + 4 + 4 + t_max * 29 + 28 + 4),
           "tables drawn on the device + one launch; the reference's host loop over the same ROIs (%d single-pair IoU calls with a "
           "device->host read each): %.1f ms" % (n_pairs, ms_h))
    # ---- the whole target layer at BASELINE config 4's per-rank shapes (2 scenes x 512 proposals -> 64 ROIs, 16384 x 128 features)
    bsz, m, n = 2, 512, 16384
    rl, gl = [], []
    for i in range(bsz):
        bx, _ = synth.proposal_boxes(m + 12, seed=200 + i, num_objects=12, jitter=0.4)
        gt = torch.zeros((20, 7)); gt[:12] = bx[m:]
        rl.append(bx[:m]); gl.append(gt)
    layer_in = {"roi_boxes3d": torch.stack(rl).to(dev), "gt_boxes3d": torch.stack(gl).to(dev),
                "rpn_xyz": synth.scenes("kitti", bsz, n, seed=9).to(dev), "rpn_features": torch.randn((bsz, n, 128), generator=g).to(dev),
                "seg_mask": (torch.rand((bsz, n), generator=g) > 0.5).float().to(dev), "pts_depth": (torch.rand((bsz, n), generator=g) * 70).to(dev)}
    layer = ptl.ProposalTargetLayer()
    ms = timeit(lambda: layer(layer_in))
    report("ProposalTargetLayer.forward", {"B": bsz, "proposals": m, "rois": 64, "N": n, "C": 128}, ms,
           bsz * (n * 12 + n * 130 * 4 + 64 * 28 + 64 * 512 * 133 * 4 + 64 * 4),
           "IoU + sampling (2 host syncs) + batched augmentation + roipool3d + canonical transform + labels; bytes = the roipool3d figure")
    # ---- proposal layer at the training shapes: 2 scenes x 16384 points, pre 9000 / post 512, thresh 0.85
    from epnet_amd import proposal_layer as pl
    bsz, n = 2, 16384
    xyz_p = synth.scenes("kitti", bsz, n, seed=21).to(dev)
    reg_p = (torch.randn((bsz, n, 76), generator=g) * 0.5).to(dev)
    sc_p = torch.randn((bsz, n), generator=g).to(dev)
    for mode in ("TRAIN", "TEST"):
        layer_p = pl.ProposalLayer(mode).to(dev)
        mcfg = getattr(layer_p.cfg, mode)
        props = layer_p.decode(reg_p, xyz_p).contiguous()
        cnt = torch.zeros((bsz,), dtype=i32, device=dev)
        layer_p.propose(sc_p, props, cnt)

        def host_loop():   # the reference's per-scene structure (proposal_layer.py:40-54, 58-119) on this package's NMS surface
            order = torch.sort(sc_p, dim=1, descending=True)[1]
            ret = torch.zeros((bsz, mcfg.RPN_POST_NMS_TOP_N, 7), device=dev)
            pre = [0, int(mcfg.RPN_PRE_NMS_TOP_N * 0.7), mcfg.RPN_PRE_NMS_TOP_N - int(mcfg.RPN_PRE_NMS_TOP_N * 0.7)]
            post = [0, int(mcfg.RPN_POST_NMS_TOP_N * 0.7), mcfg.RPN_POST_NMS_TOP_N - int(mcfg.RPN_POST_NMS_TOP_N * 0.7)]
            for k in range(bsz):
                s_ord, p_ord = sc_p[k][order[k]], props[k][order[k]]
                dist, rng, outs = p_ord[:, 2], [0, 40.0, 80.0], []
                for i in (1, 2):
                    m = (dist > rng[i - 1]) & (dist <= rng[i])
                    cur_s, cur_p = s_ord[m][:pre[i]], p_ord[m][:pre[i]]
                    keep = iou3d_utils.nms_normal_gpu(kitti_utils.boxes3d_to_bev_torch(cur_p), cur_s, mcfg.RPN_NMS_THRESH)[:post[i]]
                    outs.append(cur_p[keep])
                allp = torch.cat(outs, 0)
                ret[k, :allp.size(0)] = allp
            return ret
        same = torch.equal(host_loop(), layer_p.propose(sc_p, props)[0])
        ms_dec = timeit(lambda: layer_p.decode(reg_p, xyz_p))
        ms_prop = timeit(lambda: layer_p.propose(sc_p, props))
        ms_host = timeit(host_loop)
        report("ProposalLayer.propose", {"B": bsz, "N": n, "mode": mode, "pre": mcfg.RPN_PRE_NMS_TOP_N, "post": mcfg.RPN_POST_NMS_TOP_N,
                                         "thresh": mcfg.RPN_NMS_THRESH, "kept": cnt.tolist()}, ms_prop,
               bsz * (n * 28 + n * 4 + n * 8 + mcfg.RPN_POST_NMS_TOP_N * 32),
               "sort + bin compaction + batched NMS + gather, no host sync; per-scene host loop on the same NMS kernels: %.3f ms "
               "(same result: %s); box decoding (stock tensor ops): %.3f ms" % (ms_host, same, ms_dec))
    # ---- LI-Fusion point-to-pixel sampler: the image pyramid of the yaml (LI_FUSION.IMG_CHANNELS) at 2 scenes
    import torch.nn.functional as F
    from epnet_amd.li_fusion import Feature_Gather
    for (c, h, w, n) in ((64, 192, 640, 4096), (128, 96, 320, 1024), (256, 48, 160, 256), (512, 24, 80, 64), (32, 384, 1280, 16384)):
        fmap = torch.randn((2, c, h, w), generator=g).to(dev)
        xy = (torch.rand((2, n, 2), generator=g) * 2 - 1).to(dev)
        ms_stock = timeit(lambda: F.grid_sample(fmap, xy.unsqueeze(1), mode="bilinear", padding_mode="zeros", align_corners=True))
        ms = timeit(lambda: Feature_Gather(fmap, xy))
        report("Feature_Gather", {"B": 2, "C": c, "H": h, "W": w, "N": n}, ms, 2 * (n * 8 + c * n * 4 * 4 + c * n * 4),
               "bytes = 4 taps read + 1 value written per (point, channel); stock grid_sample: %.4f ms" % ms_stock)
    # ---- roipool3d: (B,16384,3)+(B,16384,130) -> (B,64,512,133)
    for bsz, m in ((2, 64), (1, 100), (16, 64)):
        pts = synth.scenes("kitti", bsz, 16384, seed=5).to(dev)
        feat = torch.randn((bsz, 16384, 130), generator=g).to(dev)
        boxes = torch.stack([synth.proposal_boxes(m, seed=50 + i)[0] for i in range(bsz)]).to(dev)
        pooled = torch.zeros((bsz, m, 512, 133), dtype=f32, device=dev)
        flag = torch.zeros((bsz, m), dtype=i32, device=dev)
        big = kitti_utils.enlarge_box3d(boxes.view(-1, 7), 0.2).view(bsz, m, 7).contiguous()
        ms = timeit(lambda: roipool3d_cuda.forward(pts, big, feat, pooled, flag))
        n = 16384
        report("roipool3d forward", {"B": bsz, "N": n, "M": m, "S": 512, "C": 130}, ms,
               bsz * (n * 12 + n * 130 * 4 + m * 28 + m * 512 * 133 * 4 + m * 4), "empty boxes: %d" % int(flag.sum()))
    # ---- FP ops
    for bsz, (c, m, n) in ((16, (256, 4096, 16384)), (16, (512, 1024, 4096)), (1, (256, 4096, 16384))):
        unknown = synth.scenes("kitti", bsz, n, seed=7).to(dev)
        # as in an FP module: the known set is the FPS subset of the unknown one
        kidx = torch.empty((bsz, m), dtype=i32, device=dev)
        known = torch.empty((bsz, m, 3), device=dev)
        p2.sample_centres_wrapper(bsz, n, m, unknown, p2.scene_index(unknown), kidx, known)
        d2 = torch.empty((bsz, n, 3), device=dev); idx = torch.empty((bsz, n, 3), dtype=i32, device=dev)
        ms = timeit(lambda: p2.three_nn_wrapper(bsz, n, m, unknown, known, d2, idx))
        report("three_nn", {"B": bsz, "n": n, "m": m}, ms, bsz * (n * 12 + m * 12 + n * 24), "builds its own index of the known set")
        ui, ki = p2.scene_index(unknown), p2.scene_index(known)
        if ki is not None:
            ms = timeit(lambda: p2.three_nn_indexed_wrapper(bsz, n, m, unknown, known, ui, ki, d2, idx))
            report("three_nn", {"B": bsz, "n": n, "m": m}, ms, bsz * (n * 12 + m * 12 + n * 24), "over the scene indices of both point sets")
        feats = torch.randn((bsz, c, m), generator=g).to(dev)
        w = torch.rand((bsz, n, 3), generator=g).to(dev); w = (w / w.sum(-1, keepdim=True)).contiguous()
        out = torch.empty((bsz, c, n), device=dev)
        ms = timeit(lambda: p2.three_interpolate_wrapper(bsz, c, m, n, feats, idx, w, out))
        report("three_interpolate", {"B": bsz, "C": c, "m": m, "n": n}, ms, bsz * (c * m * 4 + n * 24 + c * n * 4))
        go = torch.randn((bsz, c, n), generator=g).to(dev); gp = torch.zeros((bsz, c, m), device=dev)
        ms = timeit(lambda: p2.three_interpolate_grad_wrapper(bsz, c, n, m, go, idx, w, gp))
        report("three_interpolate_grad", {"B": bsz, "C": c, "n": n, "m": m}, ms, bsz * (c * n * 4 + n * 24 + c * m * 4))
    # ---- grouping gradient (level 2: C=96, N=4096, M=1024, ns=32) on real ball-query neighbour lists
    for bsz in (16,):
        pts = synth.scenes("kitti", bsz, 4096, seed=11).to(dev)
        cidx = torch.empty((bsz, 1024), dtype=i32, device=dev)
        ctr = torch.empty((bsz, 1024, 3), device=dev)
        p2.sample_centres_wrapper(bsz, 4096, 1024, pts, p2.scene_index(pts), cidx, ctr)
        idx = torch.empty((bsz, 1024, 32), dtype=i32, device=dev)
        p2.ball_query_wrapper(bsz, 4096, 1024, 1.0, 32, ctr, pts, idx)
        go = torch.randn((bsz, 96, 1024, 32), generator=g).to(dev)
        gp = torch.zeros((bsz, 96, 4096), device=dev)
        ms = timeit(lambda: p2.group_points_grad_wrapper(bsz, 96, 4096, 1024, 32, go, idx, gp))
        report("group_points_grad", {"B": bsz, "C": 96, "N": 4096, "M": 1024, "ns": 32}, ms, bsz * (96 * 1024 * 32 * 4 + 1024 * 32 * 4 + 96 * 4096 * 4),
               "neighbour lists of ball_query(r=1.0) around FPS centres")
    # ---- BASELINE config 5: dense 65536-point scenes, one SA level with nsample = 64 (ball-query stress)
    for bsz in (1, 16):
        n, m, ns, c, radius = 65536, 16384, 64, 64, 0.5
        xyz = synth.scenes("kitti", bsz, n, seed=9).to(dev)
        index = torch.empty((p2.scene_index_bytes(bsz, n),), dtype=torch.uint8, device=dev)
        ms = timeit(lambda: p2.scene_index_build_wrapper(bsz, n, xyz, index))
        report("scene_index_build", {"B": bsz, "N": n}, ms, bsz * (n * 12 + n * 16), "this implementation's own structure")
        temp = torch.empty((bsz, n), device=dev); fidx = torch.empty((bsz, m), dtype=i32, device=dev)

        def fps():
            temp.fill_(1e10)
            p2.furthest_point_sampling_indexed_wrapper(bsz, n, m, xyz, index, temp, fidx)
        ms = timeit(fps) if bsz == 1 else timeit(fps)
        report("furthest_point_sampling", {"B": bsz, "N": n, "M": m}, ms, bsz * (n * 12 + m * 4), "big-scene kernel over the index")
        new_xyz = torch.gather(xyz, 1, fidx.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
        bq = torch.empty((bsz, m, ns), dtype=i32, device=dev)
        ms = timeit(lambda: p2.ball_query_indexed_wrapper(bsz, n, m, radius, ns, new_xyz, xyz, index, bq))
        report("ball_query", {"B": bsz, "N": n, "M": m, "r": radius, "ns": ns}, ms, bsz * (n * 12 + m * 12 + m * ns * 4))
        feats = torch.randn((bsz, c, n), generator=g).to(dev)
        grouped = torch.empty((bsz, 3 + c, m, ns), device=dev)
        ms = timeit(lambda: p2.group_concat_wrapper(bsz, c, n, m, ns, xyz, new_xyz, feats, bq, grouped, True))
        report("group_concat", {"B": bsz, "C": c, "N": n, "M": m, "ns": ns}, ms,
               bsz * (2 * m * ns * 4 + 3 * n * 4 + c * n * 4 + (3 + c) * m * ns * 4))
        gx = torch.empty((bsz, 3, m, ns), device=dev)
        ms = timeit(lambda: p2.group_concat_wrapper(bsz, 0, n, m, ns, xyz, new_xyz, None, bq, gx, True))
        report("group_concat", {"B": bsz, "C": 0, "N": n, "M": m, "ns": ns}, ms, bsz * (m * ns * 4 + 3 * n * 4 + 3 * m * ns * 4))


if __name__ == "__main__":
    main()
