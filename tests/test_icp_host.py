"""CPU tests of the host-side ICP logic (closed-form fit from the 18 sums) and of the ICP oracle."""
import importlib

import numpy as np

from helpers import PKG
from oracle import icp_ref as OI


def test_umeyama_from_sums_recovers_known_transform():
    icp = importlib.import_module(PKG + ".icp")
    src, tgt, T_true, pick = OI.synthetic_pair(n_tgt=3000, n_src=2000)
    sums = OI.pair_sums(src, tgt, pick.astype(np.uint32))           # exact correspondences
    T = icp.umeyama_from_sums(sums)
    np.testing.assert_allclose(T, T_true, rtol=0, atol=5e-5)
    T2 = OI.umeyama(src.astype(np.float64), tgt[pick].astype(np.float64))
    np.testing.assert_allclose(T, T2, rtol=0, atol=1e-9)
    Tr = icp.umeyama_from_sums(sums, with_scale=False)
    np.testing.assert_allclose(np.linalg.det(Tr[:3, :3]), 1.0, atol=1e-12)


def test_umeyama_reflection_guard():
    icp = importlib.import_module(PKG + ".icp")
    rng = np.random.default_rng(0)
    p = rng.normal(size=(50, 3)).astype(np.float32)
    q = p.copy()
    q[:, 2] *= -1                                                     # a mirror image: best ROTATION, not reflection
    sums = OI.pair_sums(p, q, np.arange(50, dtype=np.uint32))
    T = icp.umeyama_from_sums(sums)
    assert np.linalg.det(T[:3, :3]) > 0


def test_oracle_nn_against_kdtree():
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(5)
    src = (rng.random((700, 3)) * 20).astype(np.float32)
    tgt = (rng.random((3000, 3)) * 20).astype(np.float32)
    idx, d2 = OI.nearest_neighbours(src, tgt)
    dist, kidx = cKDTree(tgt.astype(np.float64)).query(src.astype(np.float64))
    np.testing.assert_allclose(d2, dist ** 2, rtol=1e-5)
    assert (idx == kidx).mean() > 0.999


def test_library_umeyama_equals_numpy_svd_on_full_rank_and_planar_data():
    """r3d_umeyama_from_sums (one-sided Jacobi, the code the device-side solve runs) vs the oracle's numpy SVD."""
    icp = importlib.import_module(PKG + ".icp")
    rng = np.random.default_rng(0)
    worst = 0.0
    for trial in range(120):
        p = rng.normal(size=(40, 3)) * rng.uniform(0.1, 10)
        if trial % 5 == 0:
            p[:, 2] = 0.3                                             # planar source: rank-2 covariance
        _s, _t, T, _ = OI.synthetic_pair(n_tgt=8, n_src=4, seed=trial, s=rng.uniform(0.3, 3), angle_deg=rng.uniform(0, 179))
        q = p @ T[:3, :3].T + T[:3, 3] + rng.normal(size=p.shape) * 1e-3
        sums = OI.pair_sums(p.astype(np.float32), q.astype(np.float32), np.arange(40))
        for ws in (True, False):
            worst = max(worst, np.abs(icp.umeyama_from_sums(sums, ws) - OI.umeyama_from_sums(sums, ws)).max())
    assert worst < 1e-12, worst


def test_library_umeyama_rejects_degenerate_sums():
    icp = importlib.import_module(PKG + ".icp")
    import pytest
    with pytest.raises(ValueError):
        icp.umeyama_from_sums(np.zeros(18))
    one_point = OI.pair_sums(np.ones((5, 3), np.float32), np.ones((5, 3), np.float32), np.arange(5))
    with pytest.raises(ValueError):
        icp.umeyama_from_sums(one_point)                             # no spread in p


def test_weighted_sums_and_swap():
    rng = np.random.default_rng(2)
    p = rng.normal(size=(200, 3)).astype(np.float32)
    q = rng.normal(size=(300, 3)).astype(np.float32)
    idx, d2 = OI.nearest_neighbours(p, q)
    plain = OI.pair_sums(p, q, idx, d2)
    np.testing.assert_array_equal(plain, OI.pair_sums(p, q, idx))
    w = OI.pair_weights(d2, 0.3)
    assert (w == 0).any() and (w > 0).any() and w.max() < 1
    ws = OI.pair_sums(p, q, idx, d2, -1.0, 0.3)
    np.testing.assert_allclose(ws[0], w.sum(), rtol=1e-14)
    # exchanging the roles by hand == swap_pair_sums
    qq = q[idx.astype(np.int64)]
    by_hand = OI.pair_sums(qq, p, np.arange(200), d2, -1.0, 0.3)
    np.testing.assert_allclose(OI.swap_pair_sums(ws), by_hand, rtol=1e-13, atol=1e-13)
    icp = importlib.import_module(PKG + ".icp")
    np.testing.assert_array_equal(icp.swap_pair_sums(ws), OI.swap_pair_sums(ws))


def test_oracle_full_estimator_recovers_survey_c3_recipe_small():
    """SURVEY C3 recipe (s=1.7, 10 degrees, |t|=0.5) from identity, at a size the brute-force oracle finishes in seconds."""
    src, tgt, T_true, _ = OI.synthetic_pair(n_tgt=3000, n_src=3000, s=1.7, angle_deg=10.0, t_norm=0.5, seed=7)
    T, info = OI.icp_similarity_auto(src, tgt)
    np.testing.assert_allclose(T, T_true, rtol=0, atol=1e-4)
    # plain ICP from identity does NOT get there: the scale gap is outside its basin
    assert np.abs(OI.icp_similarity(src, tgt, max_iter=30) - T_true).max() > 0.1
