"""CPU: the host-callable pieces of the point-to-plane registration path (no GPU): the library's 6x6 step against the
oracle's, the oracle loop against known relative poses and an independent SciPy statement, raster-shape inference of a
camera txt, and the baseline-ratio scale (readme.md:25)."""
import importlib
import os
import sys

import numpy as np
import pytest

from helpers import PKG, ROOT
from oracle import fusion_ref as O
from oracle import plane_ref as PR

icp = importlib.import_module(PKG + ".icp")
S = importlib.import_module(PKG + ".synthetic")


def rot(axis, deg):
    a = np.deg2rad(deg)
    axis = np.asarray(axis, float) / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * K @ K


def kd_nn(tgt):
    from scipy.spatial import cKDTree
    tree = cKDTree(np.asarray(tgt, dtype=np.float64))

    def nn(cur, _tgt):
        d, j = tree.query(np.asarray(cur, dtype=np.float64))
        return j.astype(np.uint32), (d * d).astype(np.float32)
    return nn


def two(h, w, yaw=15.0, baseline=(0.35, 0.05, -0.2), noise=0.0):
    v = S.two_views(h, w, yaw_deg=yaw, baseline=baseline, depth_noise=noise, seed=1)
    pa = O.unproject(v["depth_a"], *v["K"]).astype(np.float32)
    pb = O.unproject(v["depth_b"], *v["K"]).astype(np.float32)
    return v, pa, pb


def test_library_step_equals_oracle_step_on_random_pairs():
    rng = np.random.default_rng(0)
    for trial in range(20):
        m = 400
        p = rng.normal(size=(m, 3)) * rng.uniform(0.5, 30)
        n = rng.normal(size=(m, 3))
        n /= np.linalg.norm(n, axis=1, keepdims=True)
        x_true = np.concatenate([rng.normal(size=3) * 0.02, rng.normal(size=3) * 0.1])
        J = np.concatenate([np.cross(p, n), n], axis=1)
        r = -(J @ x_true) + rng.normal(size=m) * 1e-3
        s = np.zeros(29)
        s[0], s[1], s[2:8] = m, (r * r).sum(), (J * r[:, None]).sum(0)
        s[8:] = (J.T @ J)[np.triu_indices(6)]
        T_lib, rms_lib = icp.plane_step_from_sums(s)
        T_ref, rms_ref = PR.step_from_sums(s)
        np.testing.assert_allclose(T_lib, T_ref, atol=1e-11)
        assert abs(rms_lib - rms_ref) < 1e-14
        np.testing.assert_allclose(T_lib[:3, :3] @ T_lib[:3, :3].T, np.eye(3), atol=1e-14)
        assert np.abs(T_lib[:3, 3] - x_true[3:]).max() < 1e-2


def test_unconstrained_freedoms_are_refused_by_both():
    rng = np.random.default_rng(1)
    p = rng.normal(size=(300, 3))
    for normals in (np.tile([0.0, 0.0, 1.0], (300, 1)),                                     # one plane
                    np.where(rng.random((300, 1)) < 0.5, [0.0, 0.0, 1.0], [0.0, 0.0, -1.0])):   # two parallel walls
        J = np.concatenate([np.cross(p, normals), normals], axis=1)
        s = np.zeros(29)
        s[0], s[1], s[2:8] = 300, 1.0, J.sum(0) * 0.01
        s[8:] = (J.T @ J)[np.triu_indices(6)]
        with pytest.raises(ValueError):
            icp.plane_step_from_sums(s)
        with pytest.raises(ValueError):
            PR.step_from_sums(s)
    with pytest.raises(ValueError):
        icp.plane_step_from_sums(np.zeros(29))                                              # no pairs at all


def test_oracle_loop_recovers_the_relative_pose_of_two_views():
    h, w = 120, 160
    for yaw, base in ((12.0, (0.25, 0.03, -0.1)), (20.0, (0.2, 0.02, -0.1))):
        v, pa, pb = two(h, w, yaw, base, noise=0.001)
        E = np.eye(4)
        E[:3, :3] = rot([0.3, 1.0, 0.2], 5.0)
        E[:3, 3] = (0.06, -0.05, 0.06)
        na = PR.organized_normals(pa, h, w, 0.05)
        T, hist = PR.icp_point_to_plane(pb, pa, na, T0=E @ v["T_ab"], max_iter=25, nn=kd_nn(pa))
        assert np.abs(T - v["T_ab"]).max() < 1.5e-3, np.abs(T - v["T_ab"]).max()
        assert hist[-1] < 0.2 * hist[0]


def test_normal_classes_and_lower_quantile():
    n = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, 0, -1], [0.5, 0.5, 0.1], [0.5, -0.6, 0.6], [0.2, 0.3, 0.3],
                  [-0.3, 0.3, -0.3]], np.float32)
    assert PR.normal_classes(n).tolist() == [0, 4, 8, 20, 0, 8 + 4, 8, 4 + 1]
    v = np.array([3, 1, np.inf, 2, np.nan, 5, 4], np.float32)
    assert PR.quantile_lower(v, 0.5) == (3.0, 5) and PR.quantile_lower(v, 0.0) == (1.0, 5) and PR.quantile_lower(v, 1.0) == (5.0, 5)
    assert PR.quantile_lower(v, 0.74) == (3.0, 5) and PR.quantile_lower(v, 0.75) == (4.0, 5)
    assert PR.quantile_lower(np.array([np.nan, np.inf], np.float32), 0.5) == (np.inf, 0)


def test_raster_shape_is_read_off_a_camera_txt_cloud():
    T = importlib.import_module(PKG + ".other_tools.transfer_T_icp")
    for h, w in ((48, 64), (30, 100)):
        v, pa, _pb = two(h, w)
        assert T.infer_raster_shape(pa) == (h, w)
        holes = pa.copy()
        holes.reshape(h, w, 3)[:, 0] = 0.0                    # whole first column without depth
        holes.reshape(h, w, 3)[5:9, 10:30] = 0.0
        assert T.infer_raster_shape(holes) == (h, w)
    with pytest.raises(ValueError):
        T.infer_raster_shape(np.random.default_rng(0).normal(size=(1000, 3)) + [0, 0, 5])


def test_scale_is_the_ratio_of_the_two_baselines():
    v, _pa, _pb = two(24, 32)
    for s_true in (0.4, 1.0, 7.5):
        pose_a = (v["pose_a"][0] * 3.0, v["pose_a"][1] / s_true)          # quaternion length is irrelevant (normalised like SciPy)
        pose_b = (v["pose_b"][0], v["pose_b"][1] / s_true)
        s, T_rel = icp.scale_from_baselines(v["T_ab"], pose_a, pose_b)
        assert abs(s - s_true) < 1e-12 * s_true
        np.testing.assert_allclose(T_rel[:3, :3], v["T_ab"][:3, :3], atol=1e-12)
        np.testing.assert_allclose(T_rel[:3, 3] * s_true, v["T_ab"][:3, 3], atol=1e-12)
        np.testing.assert_allclose(T_rel, PR.relative_pose(pose_a, pose_b), atol=1e-12)
    with pytest.raises(ValueError):
        icp.scale_from_baselines(v["T_ab"], v["pose_a"], v["pose_a"])
