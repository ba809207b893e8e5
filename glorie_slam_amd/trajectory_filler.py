"""PoseTrajectoryFiller (scope row N3): same interface as /root/reference/src/trajectory_filler.py - poses of the
frames that did not become keyframes, from a linear interpolation in the Lie algebra between the neighbouring
keyframes (trajectory_filler.py:48-60) refined by 12 motion-only BA-update iterations on edges to those two
keyframes (:69-76).  Returns SE3 batches (`lie.SE3`: `.inv().matrix()` like slam.py:176-180 expects)."""
import torch

from . import lie
from .factor_graph import FactorGraph


class PoseTrajectoryFiller:
    def __init__(self, net, video, printer=None, device='cuda:0', batch=16, iters=12):
        self.cnet, self.fnet, self.update = net.cnet, net.fnet, net.update
        self.count = 0
        self.video = video
        self.device = device
        self.printer = printer
        self.batch, self.iters = batch, iters
        self.MEAN = torch.tensor([0.485, 0.456, 0.406], device=device)[:, None, None]
        self.STDV = torch.tensor([0.229, 0.224, 0.225], device=device)[:, None, None]

    def _feature_encoder(self, image):
        with torch.autocast("cuda", enabled=str(self.device).startswith("cuda")):
            return self.fnet(image)

    def interpolate(self, timestamps):
        """(Gs, t0, t1): constant-velocity poses at `timestamps` between keyframes t0 <= t < t1"""
        N = self.video.counter.value
        tt = torch.as_tensor(timestamps, device=self.device, dtype=torch.float32)
        ts = self.video.timestamp[:N]
        Ps = lie.SE3(self.video.poses[:N])
        t0 = torch.as_tensor([int((ts <= t).sum()) - 1 for t in tt.tolist()], device=self.device).clamp(min=0)
        t1 = torch.where(t0 < N - 1, t0 + 1, t0)
        dt = ts[t1] - ts[t0] + 1e-3
        dP = Ps[t1] * Ps[t0].inv()
        v = dP.log() / dt.unsqueeze(-1)
        w = v * (tt - ts[t0]).unsqueeze(-1)
        return lie.SE3.exp(w) * Ps[t0], t0, t1

    def _fill(self, timestamps, images, intrinsics):
        images = torch.stack(images, 0)
        intrinsics = torch.stack(intrinsics, 0).to(self.device)
        inputs = images.to(self.device).float()
        N, M = self.video.counter.value, len(timestamps)
        if N + M > self.video.poses.shape[0]:
            raise RuntimeError(f"PoseTrajectoryFiller: the video buffer holds {self.video.poses.shape[0]} frames, "
                               f"{N} keyframes + {M} frames to fill do not fit")
        Gs, t0, t1 = self.interpolate(timestamps)
        inputs = inputs.sub(self.MEAN).div(self.STDV)
        fmap = self._feature_encoder(inputs[None] if inputs.dim() == 4 else inputs)
        fmap = fmap.reshape(M, 1, *fmap.shape[-3:])
        # the frames are parked behind the keyframes for the optimisation
        tt = torch.as_tensor(timestamps, device=self.device, dtype=torch.float32)
        self.video.counter.value += M
        self.video[N:N + M] = (tt, images.reshape(M, *images.shape[-3:]), Gs.data, 1, None, intrinsics / 8.0,
                               fmap.to(self.video.fmaps.dtype))
        graph = FactorGraph(self.video, self.update, device=self.device)
        new = torch.arange(N, N + M, device=self.device)
        graph.add_factors(t0, new)
        graph.add_factors(t1, new)
        for _ in range(self.iters):
            graph.update(N, N + M, motion_only=True)
        out = lie.SE3(self.video.poses[N:N + M].clone())
        self.video.counter.value -= M
        return [out]

    @torch.no_grad()
    def __call__(self, image_stream):
        """image_stream: iterable of (timestamp, image [1,3,H,W], depth, intrinsic-or-None) with get_intrinsic()"""
        pose_list, timestamps, images, intrinsics = [], [], [], []
        intrinsic = image_stream.get_intrinsic()
        for item in image_stream:
            timestamps.append(float(item[0]))
            images.append(item[1])
            intrinsics.append(torch.as_tensor(intrinsic))
            if len(timestamps) == self.batch:
                pose_list += self._fill(timestamps, images, intrinsics)
                timestamps, images, intrinsics = [], [], []
        if timestamps:
            pose_list += self._fill(timestamps, images, intrinsics)
        return lie.cat(pose_list, dim=0)
