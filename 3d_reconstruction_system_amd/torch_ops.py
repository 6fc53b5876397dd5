"""Torch-facing ops over the C ABI (f4 of SURVEY.md 8(f)): `BackprojectDepth`, the layer the reference's trainer builds per
scale (monodepth2/trainer.py:150-160) and calls as `self.backproject_depth[s](depth, inputs[("inv_K", s)])`
(trainer.py:387-388).  The layer itself lives in upstream monodepth2's layers.py, which the reference does not vendor; its
definition is restated in csrc/r3d_backproject.hip.  Same constructor, same call, same [B, 4, H*W] fp32 result; the
backward pass (with respect to depth -- inv_K is data in that trainer) is a second HIP kernel.
`Project3D`, its partner in the same lines (trainer.py:158-159, 389-390: `self.project_3d[s](cam_points, K, T)`), maps the
points to grid_sample coordinates [B, H, W, 2]; its backward yields the gradients of the points AND of K and T (the pose
network trains through T): the per-pixel part and the twelve per-image sums are HIP, the 4x4 product K @ T is torch's.

torch only lends tensors and the current stream; the arithmetic is the library's.  CUDA/ROCm tensors only: there is no
CPU fallback.
"""
import torch

from . import _lib as L
from .device import Context

_ctx_cache = {}


def _ctx_for(device):
    idx = device.index if device.index is not None else torch.cuda.current_device()
    stream = torch.cuda.current_stream(device).cuda_stream
    key = (idx, stream)
    ctx = _ctx_cache.get(key)
    if ctx is None or ctx.handle is None:
        ctx = Context(idx, stream=stream)
        _ctx_cache[key] = ctx
    return ctx


def _check(depth, inv_K, batch, height, width):
    if depth.device.type != "cuda":
        raise RuntimeError("BackprojectDepth runs on an MI355X (got a tensor on %s); there is no CPU fallback" % depth.device)
    if depth.dtype != torch.float32 or inv_K.dtype != torch.float32:
        raise TypeError("BackprojectDepth is fp32 like the upstream layer")
    if depth.numel() != batch * height * width:
        raise ValueError("depth has %d elements, layer was built for %d x %d x %d" % (depth.numel(), batch, height, width))
    if tuple(inv_K.shape) != (batch, 4, 4):
        raise ValueError("inv_K must be [%d, 4, 4]" % batch)


class _BackprojectFn(torch.autograd.Function):
    @staticmethod
    def forward(fn, depth, inv_K, batch, height, width):
        _check(depth, inv_K, batch, height, width)
        d, k = depth.contiguous(), inv_K.contiguous()
        out = torch.empty((batch, 4, height * width), dtype=torch.float32, device=depth.device)
        ctx = _ctx_for(depth.device)
        L.check(ctx.lib.r3d_backproject_depth_f32(ctx.handle, d.data_ptr(), k.data_ptr(), batch, height, width, out.data_ptr()))
        fn.save_for_backward(k)
        fn.shape = (batch, height, width, tuple(depth.shape))
        return out

    @staticmethod
    def backward(fn, grad_out):
        (k,) = fn.saved_tensors
        batch, height, width, depth_shape = fn.shape
        g = grad_out.contiguous()
        grad_depth = torch.empty((batch, height * width), dtype=torch.float32, device=g.device)
        ctx = _ctx_for(g.device)
        L.check(ctx.lib.r3d_backproject_depth_grad_f32(ctx.handle, g.data_ptr(), k.data_ptr(), batch, height, width,
                                                       grad_depth.data_ptr()))
        return grad_depth.view(depth_shape), None, None, None, None


class BackprojectDepth(torch.nn.Module):
    """Layer to transform a depth image into a point cloud (upstream monodepth2 layers.BackprojectDepth)."""

    def __init__(self, batch_size, height, width):
        super().__init__()
        self.batch_size, self.height, self.width = int(batch_size), int(height), int(width)

    def forward(self, depth, inv_K):
        return _BackprojectFn.apply(depth, inv_K, self.batch_size, self.height, self.width)


class _Project3DFn(torch.autograd.Function):
    @staticmethod
    def forward(fn, points, P, batch, height, width, eps):
        if points.device.type != "cuda":
            raise RuntimeError("Project3D runs on an MI355X (got a tensor on %s); there is no CPU fallback" % points.device)
        if points.dtype != torch.float32 or P.dtype != torch.float32:
            raise TypeError("Project3D is fp32 like the upstream layer")
        if tuple(points.shape) != (batch, 4, height * width):
            raise ValueError("points must be [%d, 4, %d]" % (batch, height * width))
        if tuple(P.shape) != (batch, 3, 4):
            raise ValueError("K @ T must be [%d, 4, 4]" % batch)
        x, m = points.contiguous(), P.contiguous()
        pix = torch.empty((batch, height, width, 2), dtype=torch.float32, device=points.device)
        ctx = _ctx_for(points.device)
        L.check(ctx.lib.r3d_project3d_f32(ctx.handle, x.data_ptr(), m.data_ptr(), batch, height, width, eps, pix.data_ptr()))
        fn.save_for_backward(x, m)
        fn.geom = (batch, height, width, eps)
        return pix

    @staticmethod
    def backward(fn, grad_pix):
        x, m = fn.saved_tensors
        batch, height, width, eps = fn.geom
        g = grad_pix.contiguous()
        want_x, want_P = fn.needs_input_grad[0], fn.needs_input_grad[1]
        grad_x = torch.empty_like(x) if want_x else None
        grad_P = torch.empty_like(m) if want_P else None
        ctx = _ctx_for(g.device)
        L.check(ctx.lib.r3d_project3d_grad_f32(ctx.handle, g.data_ptr(), x.data_ptr(), m.data_ptr(), batch, height, width, eps,
                                               grad_x.data_ptr() if want_x else None,
                                               grad_P.data_ptr() if want_P else None))
        return grad_x, grad_P, None, None, None, None


class Project3D(torch.nn.Module):
    """Layer which projects 3D points into a camera with intrinsics K and at position T (upstream monodepth2
    layers.Project3D): [B, 4, H*W] points -> [B, H, W, 2] sampling coordinates in [-1, 1]."""

    def __init__(self, batch_size, height, width, eps=1e-7):
        super().__init__()
        self.batch_size, self.height, self.width, self.eps = int(batch_size), int(height), int(width), float(eps)

    def forward(self, points, K, T):
        P = torch.matmul(K, T)[:, :3, :]
        return _Project3DFn.apply(points, P, self.batch_size, self.height, self.width, self.eps)
