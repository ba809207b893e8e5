"""The subset of lietorch's SE3 the drivers need (lietorch v0.2, /root/reference/src/trajectory_filler.py:48-60,
motion_filter.py:50, depth_video.py:314-315): group product, inverse, exp / log, matrix, on [..., 7] tensors
`[tx ty tz qx qy qz qw]`, tangent `[tau, phi]` - the conventions of the reference's device helpers
(/root/reference/src/lib/droid_kernels.cu:58-175).  Plain torch on a handful of poses: host-side bookkeeping, not a
kernel."""
import torch


def _qmul(a, b):
    ax, ay, az, aw = a.unbind(-1)
    bx, by, bz, bw = b.unbind(-1)
    return torch.stack([aw * bx + ax * bw + ay * bz - az * by, aw * by + ay * bw + az * bx - ax * bz,
                        aw * bz + az * bw + ax * by - ay * bx, aw * bw - ax * bx - ay * by - az * bz], -1)


def _qrot(q, v):
    qv = q[..., :3]
    uv = 2.0 * torch.cross(qv, v, dim=-1)
    return v + q[..., 3:4] * uv + torch.cross(qv, uv, dim=-1)


def _qconj(q):
    return q * torch.tensor([-1.0, -1.0, -1.0, 1.0], dtype=q.dtype, device=q.device)


def _hat(p):
    z = torch.zeros_like(p[..., 0])
    return torch.stack([torch.stack([z, -p[..., 2], p[..., 1]], -1), torch.stack([p[..., 2], z, -p[..., 0]], -1),
                        torch.stack([-p[..., 1], p[..., 0], z], -1)], -2)


class SE3:
    def __init__(self, data):
        self.data = data

    @staticmethod
    def Identity(*shape, device=None, dtype=torch.float32):
        d = torch.zeros(*shape, 7, device=device, dtype=dtype)
        d[..., 6] = 1.0
        return SE3(d)

    def __getitem__(self, idx):
        return SE3(self.data[idx])

    @property
    def shape(self):
        return self.data.shape[:-1]

    def vec(self):
        return self.data

    def inv(self):
        qi = _qconj(self.data[..., 3:])
        return SE3(torch.cat([-_qrot(qi, self.data[..., :3]), qi], -1))

    manifold_dim = 6

    def act(self, X):
        """homogeneous points [..., 4]: (R X[:3] + t X[3], X[3])  (droid_kernels.cu:72-79)"""
        Y = _qrot(self.data[..., 3:], X[..., :3]) + self.data[..., :3] * X[..., 3:4]
        return torch.cat([Y, X[..., 3:4]], -1)

    def adjT(self, a):
        """dual adjoint on covectors [..., 6] = [tau, phi] (droid_kernels.cu:81-97): Ad(T)^T a"""
        t, qi = self.data[..., :3], _qconj(self.data[..., 3:])
        u = torch.cross(a[..., :3], t.expand_as(a[..., :3]), dim=-1)
        return torch.cat([_qrot(qi, a[..., :3]), _qrot(qi, a[..., 3:]) + _qrot(qi, u)], -1)

    def retr(self, a):
        """left retraction exp(a) * T (droid_kernels.cu:877-896)"""
        return SE3.exp(a) * self

    def __mul__(self, other):
        if isinstance(other, torch.Tensor):
            return self.act(other)
        q = _qmul(self.data[..., 3:], other.data[..., 3:])
        t = _qrot(self.data[..., 3:], other.data[..., :3]) + self.data[..., :3]
        return SE3(torch.cat([t, q], -1))

    def matrix(self):
        x, y, z, w = self.data[..., 3:].unbind(-1)
        R = torch.stack([torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], -1),
                         torch.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], -1),
                         torch.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1)], -2)
        T = torch.zeros(*self.data.shape[:-1], 4, 4, dtype=self.data.dtype, device=self.data.device)
        T[..., :3, :3] = R
        T[..., :3, 3] = self.data[..., :3]
        T[..., 3, 3] = 1.0
        return T

    @staticmethod
    def exp(xi):
        tau, phi = xi[..., :3], xi[..., 3:]
        th2 = (phi * phi).sum(-1, keepdim=True)
        th = th2.sqrt()
        small = th2 < 1e-8
        ths = torch.where(small, torch.ones_like(th), th)
        imag = torch.where(small, 0.5 - th2 / 48.0 + th2 * th2 / 3840.0, torch.sin(0.5 * ths) / ths)
        real = torch.where(small, 1.0 - th2 / 8.0 + th2 * th2 / 384.0, torch.cos(0.5 * ths))
        q = torch.cat([imag * phi, real], -1)
        big = th > 1e-4
        th2s = torch.where(big, th2, torch.ones_like(th2))
        a = torch.where(big, (1.0 - torch.cos(ths)) / th2s, torch.zeros_like(th))
        b = torch.where(big, (ths - torch.sin(ths)) / (ths * th2s), torch.zeros_like(th))
        c1 = torch.cross(phi, tau, dim=-1)
        c2 = torch.cross(phi, c1, dim=-1)
        return SE3(torch.cat([tau + a * c1 + b * c2, q], -1))

    def log(self):
        t, q = self.data[..., :3], self.data[..., 3:]
        sq = (q[..., :3] * q[..., :3]).sum(-1, keepdim=True)
        n = sq.sqrt()
        w = q[..., 3:4]
        # phi = 2 atan2(|qv|, w) qv / |qv|  (Taylor around 0; w < 0 handled through the sign like lietorch)
        small = sq < 1e-12
        ns = torch.where(small, torch.ones_like(n), n)
        ang = 2.0 * torch.atan2(n, w.abs()) * torch.sign(torch.where(w == 0, torch.ones_like(w), w))
        scale = torch.where(small, 2.0 / w - 2.0 * sq / (3.0 * w * w * w), ang / ns)
        phi = scale * q[..., :3]
        th2 = (phi * phi).sum(-1, keepdim=True)
        th = th2.sqrt()
        P = _hat(phi)
        big = th > 1e-4
        ths = torch.where(big, th, torch.ones_like(th))
        coef = torch.where(big, (1.0 - 0.5 * ths * torch.sin(ths) / (1.0 - torch.cos(ths)).clamp_min(1e-30)) / (ths * ths),
                           torch.full_like(th, 1.0 / 12.0))
        eye = torch.eye(3, dtype=t.dtype, device=t.device).expand(P.shape)
        Vinv = eye - 0.5 * P + coef.unsqueeze(-1) * (P @ P)
        tau = (Vinv @ t.unsqueeze(-1)).squeeze(-1)
        return torch.cat([tau, phi], -1)


def cat(items, dim=0):
    return SE3(torch.cat([s.data for s in items], dim))
