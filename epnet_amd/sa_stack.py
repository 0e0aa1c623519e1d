"""The set-abstraction (SA) / feature-propagation (FP) OPERATOR stack of the RPN backbone, without the MLPs.

This is the unit BASELINE.json's metric is quoted on ("SA-stack points/sec per GPU, 16384-pt KITTI
scene"): per scene and level, furthest_point_sample -> gather_operation -> for each of the two MSG
scales ball_query -> grouping_operation(xyz) [-> grouping_operation(features)] -- i.e. 4 FPS +
4 gather + 8 ball_query + 14 grouping calls for the pyramid 16384 -> 4096 -> 1024 -> 256 -> 64
(reference call sequence: pointnet2_modules.py:39-59 + pointnet2_utils.py:249-255, shapes from
tools/cfgs/LI_Fusion_with_attention_use_ce_loss.yaml:56-64). The shared MLPs / max-pool between the
levels are stock PyTorch (out of scope), so the feature tensor entering each level is a synthetic
input of the right shape (C = 0 / 96 / 256 / 512).

Every buffer is allocated once; ``run()`` only launches kernels (through the extension stand-ins,
i.e. through the C ABI), so the sequence can be captured into a HIP graph (``capture()``).
"""
import torch

from . import pointnet2_cuda as ext

RPN_NPOINTS = (4096, 1024, 256, 64)
RPN_RADII = ((0.1, 0.5), (0.5, 1.0), (1.0, 2.0), (2.0, 4.0))
RPN_NSAMPLES = ((16, 32), (16, 32), (16, 32), (16, 32))
RPN_FEAT_CHANNELS = (0, 96, 256, 512)  # channels of the features grouped at each level
# FP modules, deepest first: (C of known feats, m known, n unknown), pointnet2_msg.py:232-235
RPN_FP = ((1024, 64, 256), (512, 256, 1024), (512, 1024, 4096), (256, 4096, 16384))


def sa_algorithmic_bytes(n=16384, npoints=RPN_NPOINTS, nsamples=RPN_NSAMPLES, feat_channels=RPN_FEAT_CHANNELS):
    """compulsory fp32/int32 traffic of one scene through the SA op stack (each input read once, each
    output written once), per kernel family -- SURVEY.md section 8(d)"""
    out = {"fps": 0, "gather": 0, "ball_query": 0, "group_xyz": 0, "group_feat": 0}
    cur = n
    for lvl, m in enumerate(npoints):
        out["fps"] += cur * 12 + m * 4
        out["gather"] += m * 4 + 3 * cur * 4 + 3 * m * 4
        for ns in nsamples[lvl]:
            out["ball_query"] += cur * 12 + m * 12 + m * ns * 4
            out["group_xyz"] += m * ns * 4 + 3 * cur * 4 + 3 * m * ns * 4
            c = feat_channels[lvl]
            if c:
                out["group_feat"] += m * ns * 4 + c * cur * 4 + c * m * ns * 4
        cur = m
    out["total"] = sum(out.values())
    return out


def fp_algorithmic_bytes(fp=RPN_FP):
    out = {"three_nn": 0, "three_interpolate": 0}
    for c, m, n in fp:
        out["three_nn"] += n * 12 + m * 12 + n * 24
        out["three_interpolate"] += c * m * 4 + n * 24 + c * n * 4
    out["total"] = sum(out.values())
    return out


class SAStack:
    def __init__(self, batch, n=16384, device="cuda", npoints=RPN_NPOINTS, radii=RPN_RADII, nsamples=RPN_NSAMPLES,
                 feat_channels=RPN_FEAT_CHANNELS, with_fp=False, fp=RPN_FP, seed=0, overlap=True):
        self.batch, self.n = batch, n
        # overlap: FPS/gather of level l+1 depend only on the centres of level l (never on features), so
        # the sampling chain runs ahead on the launch stream while ball query + grouping of each level
        # follow on a second HIP stream (the SA levels' FPS is a latency-bound one-workgroup-per-scene
        # kernel that leaves the rest of the chip idle)
        self.overlap = overlap
        self.side = None
        self.npoints, self.radii, self.nsamples, self.feat_channels = npoints, radii, nsamples, feat_channels
        self.with_fp, self.fp = with_fp, fp
        dev = torch.device(device)
        g = torch.Generator().manual_seed(seed)
        f32, i32 = torch.float32, torch.int32
        self.levels = []
        cur = n
        for lvl, m in enumerate(npoints):
            c = feat_channels[lvl]
            L = {
                "n": cur, "m": m, "c": c,
                "xyz_t": torch.empty((batch, 3, cur), dtype=f32, device=dev),
                "temp": torch.empty((batch, cur), dtype=f32, device=dev),
                "fps_idx": torch.empty((batch, m), dtype=i32, device=dev),
                "new_xyz_t": torch.empty((batch, 3, m), dtype=f32, device=dev),
                "new_xyz": torch.empty((batch, m, 3), dtype=f32, device=dev),
                "features": (torch.randn((batch, c, cur), generator=g, dtype=f32).to(dev) if c else None),
                "scales": [],
            }
            for radius, ns in zip(radii[lvl], nsamples[lvl]):
                L["scales"].append({
                    "radius": radius, "ns": ns,
                    "idx": torch.empty((batch, m, ns), dtype=i32, device=dev),
                    "grouped_xyz": torch.empty((batch, 3, m, ns), dtype=f32, device=dev),
                    "grouped_feat": (torch.empty((batch, c, m, ns), dtype=f32, device=dev) if c else None),
                })
            self.levels.append(L)
            cur = m
        self.fp_bufs = []
        if with_fp:
            for c, m, nn_ in fp:
                self.fp_bufs.append({
                    "c": c, "m": m, "n": nn_,
                    "known_feats": torch.randn((batch, c, m), generator=g, dtype=f32).to(dev),
                    "dist2": torch.empty((batch, nn_, 3), dtype=f32, device=dev),
                    "idx": torch.empty((batch, nn_, 3), dtype=i32, device=dev),
                    "out": torch.empty((batch, c, nn_), dtype=f32, device=dev),
                })
        self.graph = None
        self.static_xyz = None

    def _group_level(self, L, cur_xyz):
        b, n, m = self.batch, L["n"], L["m"]
        for S in L["scales"]:
            ext.ball_query_wrapper(b, n, m, S["radius"], S["ns"], L["new_xyz"], cur_xyz, S["idx"])
            ext.group_points_wrapper(b, 3, n, m, S["ns"], L["xyz_t"], S["idx"], S["grouped_xyz"])
            if L["c"]:
                ext.group_points_wrapper(b, L["c"], n, m, S["ns"], L["features"], S["idx"], S["grouped_feat"])

    def run(self, xyz):
        """xyz (B,N,3) contiguous fp32 on the stack's device; all outputs land in self.levels"""
        b = self.batch
        main = torch.cuda.current_stream(xyz.device)
        if self.overlap and self.side is None:
            self.side = torch.cuda.Stream(device=xyz.device)
        cur_xyz = xyz
        for L in self.levels:
            n, m = L["n"], L["m"]
            L["xyz_t"].copy_(cur_xyz.transpose(1, 2))            # pointnet2_modules.py:30
            L["temp"].fill_(1e10)                                # pointnet2_utils.py:26
            ext.furthest_point_sampling_wrapper(b, n, m, cur_xyz, L["temp"], L["fps_idx"])
            ext.gather_points_wrapper(b, 3, n, m, L["xyz_t"], L["fps_idx"], L["new_xyz_t"])
            L["new_xyz"].copy_(L["new_xyz_t"].transpose(1, 2))   # pointnet2_modules.py:42-45
            if self.overlap:
                self.side.wait_stream(main)
                with torch.cuda.stream(self.side):
                    self._group_level(L, cur_xyz)
            else:
                self._group_level(L, cur_xyz)
            cur_xyz = L["new_xyz"]
        if self.overlap:
            main.wait_stream(self.side)
        if self.with_fp:
            # FP modules walk back up: unknown = xyz of the finer level, known = the coarser one
            xyzs = [xyz] + [L["new_xyz"] for L in self.levels]
            for k, F in enumerate(self.fp_bufs):
                known = xyzs[len(self.levels) - k]
                unknown = xyzs[len(self.levels) - k - 1]
                ext.three_nn_wrapper(b, F["n"], F["m"], unknown, known, F["dist2"], F["idx"])
                inv = 1.0 / (torch.sqrt(F["dist2"]) + 1e-8)      # pointnet2_modules.py:157-159
                weight = inv / torch.sum(inv, dim=2, keepdim=True)
                ext.three_interpolate_wrapper(b, F["c"], F["m"], F["n"], F["known_feats"], F["idx"], weight, F["out"])

    def capture(self, xyz):
        """capture run() into a HIP graph (torch.cuda.CUDAGraph); replay with self.replay()"""
        self.static_xyz = xyz.clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self.run(self.static_xyz)  # warm-up outside capture
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.run(self.static_xyz)
        return self.graph

    def replay(self, xyz=None):
        if xyz is not None:
            self.static_xyz.copy_(xyz)
        self.graph.replay()
