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
