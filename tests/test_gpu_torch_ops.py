"""GPU: torch-facing BackprojectDepth (f4) against a plain PyTorch fp32 statement of upstream monodepth2's layer
(the formula the reference's trainer relies on, monodepth2/trainer.py:150-160, 387-390), forward and backward.
Tolerance: fp32, |err| <= 1e-5 * (1 + |ref|) -- the layer's matmul has no defined summation order."""
import importlib

import numpy as np
import pytest

from helpers import PKG

pytestmark = pytest.mark.gpu


def reference_backproject(depth, inv_K, batch, height, width):
    import torch
    ys, xs = np.meshgrid(range(height), range(width), indexing="ij")          # upstream: meshgrid(range(w), range(h), 'xy')
    pix = torch.from_numpy(np.stack([xs.reshape(-1), ys.reshape(-1), np.ones(height * width)], 0).astype(np.float32))
    pix = pix.unsqueeze(0).repeat(batch, 1, 1).to(depth.device)
    cam = torch.matmul(inv_K[:, :3, :3], pix)
    cam = depth.view(batch, 1, -1) * cam
    return torch.cat([cam, torch.ones(batch, 1, height * width, device=depth.device)], 1)


@pytest.mark.parametrize("shape", [(1, 4, 6), (3, 24, 32), (2, 192, 640), (12, 96, 320), (1, 480, 640)])
def test_backproject_depth_forward_backward(shape):
    import torch
    T = importlib.import_module(PKG + ".torch_ops")
    b, h, w = shape
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(b * 1000 + h)
    depth = (torch.rand((b, 1, h, w), generator=g) * 80 + 0.1).to(dev).requires_grad_(True)
    K = torch.eye(4).repeat(b, 1, 1)
    K[:, 0, 0], K[:, 1, 1], K[:, 0, 2], K[:, 1, 2] = 0.58 * w, 1.92 * h, 0.5 * w, 0.5 * h
    K[:, 0, 1] = torch.rand(b, generator=g) * 0.01                            # a little skew so every entry matters
    inv_K = torch.linalg.inv(K).to(dev)
    layer = T.BackprojectDepth(b, h, w)
    out = layer(depth, inv_K)
    depth_ref = depth.detach().clone().requires_grad_(True)
    ref = reference_backproject(depth_ref, inv_K, b, h, w)
    assert out.shape == ref.shape == (b, 4, h * w) and out.dtype == torch.float32
    err = (out - ref).abs() / (1 + ref.abs())
    assert float(err.max()) <= 1e-5, float(err.max())
    assert torch.equal(out[:, 3], torch.ones_like(out[:, 3]))
    weight = torch.rand(out.shape, generator=g).to(dev)
    (out * weight).sum().backward()
    (ref * weight).sum().backward()
    gerr = (depth.grad - depth_ref.grad).abs() / (1 + depth_ref.grad.abs())
    assert depth.grad.shape == depth.shape and float(gerr.max()) <= 1e-5, float(gerr.max())


def test_backproject_depth_refuses_cpu_and_bad_shapes():
    import torch
    T = importlib.import_module(PKG + ".torch_ops")
    layer = T.BackprojectDepth(1, 4, 6)
    with pytest.raises(RuntimeError):
        layer(torch.ones(1, 1, 4, 6), torch.eye(4).unsqueeze(0))
    dev = torch.device("cuda", 0)
    with pytest.raises(ValueError):
        layer(torch.ones(1, 1, 4, 7, device=dev), torch.eye(4, device=dev).unsqueeze(0))
    with pytest.raises(TypeError):
        layer(torch.ones(1, 1, 4, 6, device=dev, dtype=torch.float64), torch.eye(4, device=dev).unsqueeze(0))
