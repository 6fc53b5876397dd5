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


def reference_project3d(points, K, T, batch, height, width, eps=1e-7):
    """upstream monodepth2 layers.Project3D.forward, statement for statement."""
    import torch
    P = torch.matmul(K, T)[:, :3, :]
    cam_points = torch.matmul(P, points)
    pix_coords = cam_points[:, :2, :] / (cam_points[:, 2, :].unsqueeze(1) + eps)
    pix_coords = pix_coords.view(batch, 2, height, width)
    pix_coords = pix_coords.permute(0, 2, 3, 1)
    pix_coords = pix_coords / torch.tensor([width - 1, height - 1], dtype=torch.float32, device=points.device)
    return (pix_coords - 0.5) * 2


@pytest.mark.parametrize("shape", [(1, 4, 6), (3, 24, 32), (2, 192, 640), (12, 96, 320), (1, 480, 640)])
def test_project3d_forward_backward_through_the_trainer_pair(shape):
    """depth -> BackprojectDepth -> Project3D (trainer.py:387-390) with gradients into depth, K and T.
    Tolerance: fp32; forward |err| <= 1e-5 (1 + |ref|) on coordinates of O(1); the per-image K / T gradients are sums over
    H*W pixels (fp32 partials, fixed order here, rocBLAS order in the reference): <= 2e-4 of the gradient's largest entry."""
    import torch
    TO = importlib.import_module(PKG + ".torch_ops")
    b, h, w = shape
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(b * 77 + w)
    depth0 = (torch.rand((b, 1, h, w), generator=g) * 40 + 2.0).to(dev)
    K0 = torch.eye(4).repeat(b, 1, 1)
    K0[:, 0, 0], K0[:, 1, 1], K0[:, 0, 2], K0[:, 1, 2] = 0.58 * w, 1.92 * h, 0.5 * w, 0.5 * h
    inv_K = torch.linalg.inv(K0).to(dev)
    ang = (torch.rand(b, generator=g) - 0.5) * 0.1
    T0 = torch.eye(4).repeat(b, 1, 1)
    T0[:, 0, 0], T0[:, 0, 2], T0[:, 2, 0], T0[:, 2, 2] = torch.cos(ang), torch.sin(ang), -torch.sin(ang), torch.cos(ang)
    T0[:, :3, 3] = (torch.rand((b, 3), generator=g) - 0.5) * 0.6
    weight = torch.rand((b, h, w, 2), generator=g).to(dev)
    grads = []
    for mine in (True, False):
        depth = depth0.clone().requires_grad_(True)
        K = K0.to(dev).requires_grad_(True)
        T = T0.to(dev).requires_grad_(True)
        if mine:
            pts = TO.BackprojectDepth(b, h, w)(depth, inv_K)
            pix = TO.Project3D(b, h, w)(pts, K, T)
        else:
            pts = reference_backproject(depth, inv_K, b, h, w)
            pix = reference_project3d(pts, K, T, b, h, w)
        assert pix.shape == (b, h, w, 2) and pix.dtype == torch.float32
        (pix * weight).sum().backward()
        grads.append((pix.detach(), depth.grad, K.grad, T.grad))
    (pix, gd, gK, gT), (pix_r, gd_r, gK_r, gT_r) = grads
    err = (pix - pix_r).abs() / (1 + pix_r.abs())
    assert float(err.max()) <= 1e-5, float(err.max())
    gerr = (gd - gd_r).abs() / (1e-6 + gd_r.abs().max())
    assert float(gerr.max()) <= 1e-5, float(gerr.max())
    for got, want in ((gK, gK_r), (gT, gT_r)):
        assert got.shape == want.shape == (b, 4, 4)
        assert float((got - want).abs().max()) <= 2e-4 * float(want.abs().max()), (got, want)


def test_project3d_grad_P_is_bitwise_repeatable_and_optional_outputs():
    import torch
    TO = importlib.import_module(PKG + ".torch_ops")
    b, h, w = 2, 192, 640
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(5)
    pts = torch.rand((b, 4, h * w), generator=g).to(dev) + 1.0
    K = (torch.eye(4).repeat(b, 1, 1) * 50).to(dev).requires_grad_(True)
    T = torch.eye(4).repeat(b, 1, 1).to(dev)
    layer = TO.Project3D(b, h, w)
    weight = torch.rand((b, h, w, 2), generator=g).to(dev)
    runs = []
    for _ in range(3):
        K.grad = None
        (layer(pts, K, T) * weight).sum().backward()                # points need no gradient here: only grad_P is formed
        runs.append(K.grad.clone())
    assert torch.equal(runs[0], runs[1]) and torch.equal(runs[0], runs[2])
    pts_g = pts.clone().requires_grad_(True)
    (layer(pts_g, K.detach(), T) * weight).sum().backward()          # and the other way round
    assert pts_g.grad is not None and torch.isfinite(pts_g.grad).all()
    with pytest.raises(RuntimeError):
        layer(pts.cpu(), K.detach().cpu(), T.cpu())
    with pytest.raises(ValueError):
        TO.Project3D(b, h, w + 1)(pts, K.detach(), T)
