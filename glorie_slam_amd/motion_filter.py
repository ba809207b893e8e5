"""MotionFilter (scope row N3): same interface as /root/reference/src/motion_filter.py - `track(tstamp, image,
intrinsics)` runs the feature encoder on every incoming frame and appends the frame to the DepthVideo when ONE
application of the update operator on the correlation with the last keyframe predicts a mean flow above `thresh`
(motion_filter.py:47-96).  The correlation lookup and the update operator are the HIP path (CorrBlock lookup,
FusedUpdate); the encoders are plain torch modules (they run once per frame, not per BA-update iteration)."""
import numpy as np
import torch

from .droid_net import CorrBlock, FusedUpdate
from .factor_graph import coords_grid


def load_mono_depth(idx, cfg):
    """datasets.py:10-15 of the reference: the mono-depth prior the estimator stored for frame `idx`"""
    path = f"{cfg['data']['output']}/{cfg['scene']}_priors/depths/{idx:05d}.npy"
    return torch.from_numpy(np.load(path))


class MotionFilter:
    def __init__(self, net, video, cfg, thresh=2.5, device="cuda:0", mono_depth_fn=None):
        """mono_depth_fn(tstamp, image) -> mono depth [H,W] replaces the on-disk prior / the online estimator
        (both out of scope: src/mono_estimators.py); None = load_mono_depth like the reference's offline mode"""
        self.cfg = cfg
        self.cnet, self.fnet, self.update = net.cnet, net.fnet, net.update
        self.video = video
        self.thresh = thresh
        self.device = device
        self.count = 0
        self.mono_depth_fn = mono_depth_fn
        self.MEAN = torch.as_tensor([0.485, 0.456, 0.406], device=device)[:, None, None]
        self.STDV = torch.as_tensor([0.229, 0.224, 0.225], device=device)[:, None, None]
        self._fused = FusedUpdate(self.update) if str(device).startswith("cuda") else None

    def _context_encoder(self, image):
        with torch.autocast("cuda", enabled=str(self.device).startswith("cuda")):
            net, inp = self.cnet(image).split([128, 128], dim=2)
            return net.tanh().squeeze(0), inp.relu().squeeze(0)

    def _feature_encoder(self, image):
        with torch.autocast("cuda", enabled=str(self.device).startswith("cuda")):
            return self.fnet(image).squeeze(0)

    def _mono(self, tstamp, image):
        if self.mono_depth_fn is not None:
            return self.mono_depth_fn(tstamp, image)
        return load_mono_depth(int(tstamp), self.cfg)

    @torch.no_grad()
    def track(self, tstamp, image, intrinsics=None):
        """image [1,3,H,W] in [0,1]; intrinsics [4] at full resolution"""
        s = self.video.down_scale
        ht, wd = image.shape[-2] // s, image.shape[-1] // s
        inputs = image[None].to(self.device).float().clone()
        inputs = inputs.sub_(self.MEAN).div_(self.STDV)
        gmap = self._feature_encoder(inputs)
        if self.video.counter.value == 0:
            # the first frame always becomes a keyframe (identity pose, unit disparity)
            net, inp = self._context_encoder(inputs[:, [0]])
            self.net, self.inp, self.fmap = net, inp, gmap
            ident = torch.tensor([0, 0, 0, 0, 0, 0, 1.0], device=self.device)
            self.video.append(tstamp, image[0], ident, 1.0, self._mono(tstamp, image),
                              intrinsics / float(s), gmap, net[0, 0], inp[0, 0])
            return True
        # one update iteration on the all-pairs correlation with the last keyframe at zero flow
        coords0 = coords_grid(ht, wd, device=self.device)[None, None]
        with torch.autocast("cuda", enabled=str(self.device).startswith("cuda")):
            corr = CorrBlock(self.fmap[None, [0]], gmap[None, [0]])(coords0)
        if self._fused is not None:
            _, delta, weight = self._fused(self.net[None], self.inp[None], corr)
        else:
            _, delta, weight = self.update(self.net[None], self.inp[None], corr)
        if delta.float().norm(dim=-1).mean().item() > self.thresh:
            self.count = 0
            net, inp = self._context_encoder(inputs[:, [0]])
            self.net, self.inp, self.fmap = net, inp, gmap
            self.video.append(tstamp, image[0], None, None, self._mono(tstamp, image), intrinsics / float(s),
                              gmap, net[0], inp[0])
            return True
        self.count += 1
        return False
