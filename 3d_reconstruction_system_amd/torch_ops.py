"""Torch-facing ops over the C ABI (f4 of SURVEY.md 8(f)): `BackprojectDepth`, the layer the reference's trainer builds per
scale (monodepth2/trainer.py:150-160) and calls as `self.backproject_depth[s](depth, inputs[("inv_K", s)])`
(trainer.py:387-388).  The layer itself lives in upstream monodepth2's layers.py, which the reference does not vendor; its
definition is restated in csrc/r3d_backproject.hip.  Same constructor, same call, same [B, 4, H*W] fp32 result; the
backward pass (with respect to depth -- inv_K is data in that trainer) is a second HIP kernel.

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
