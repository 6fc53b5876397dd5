"""GPU tests of the ICP estimation kernels (build-defined, parity unpinned by the reference):
HIP NN / covariance sums / full loop against oracle/icp_ref.py and known-answer recoveries."""
import numpy as np
import pytest

from helpers import r3d as _r3d
from oracle import icp_ref as OI

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def R():
    return _r3d()


@pytest.fixture(scope="module")
def icp(R):
    import importlib
    return importlib.import_module(R.__name__ + ".icp")


@pytest.fixture(scope="module")
def ctx(R):
    c = R.Context(0)
    yield c
    c.close()


def assert_nn_valid(src, tgt, idx, d2):
    """idx must attain the oracle's minimum distance; where it differs from the oracle's index the two
    candidates must be exact ties broken... the kernel's fma and the oracle's emulation can differ in
    the last bit on ~2^-29 of pairs, so allow a 1-ulp slack on d2 but demand exact equality of indices
    whenever the minimum is unique beyond that slack."""
    oi, od = OI.nearest_neighbours(src, tgt)
    np.testing.assert_allclose(d2, od, rtol=2e-7, atol=0)
    mism = np.nonzero(idx != oi)[0]
    for k in mism:  # rare: verify both are minima within an ulp
        dk = OI.pair_d2(src[k:k + 1], tgt)[0]
        assert dk[idx[k]] <= od[k] * (1 + 2e-7)
    assert len(mism) <= max(2, len(idx) // 10000)


@pytest.mark.parametrize("n,m", [(1, 1), (5, 3), (64, 1000), (300, 1024), (1000, 1025), (2500, 5000), (4097, 33)])
@pytest.mark.parametrize("S", [0, 1, 2, 4])
@pytest.mark.parametrize("culled", [False, True])
def test_nn_matches_oracle(icp, ctx, n, m, S, culled):
    rng = np.random.default_rng(n * 7 + m)
    src = (rng.random((n, 3)) * 20).astype(np.float32)
    tgt = (rng.random((m, 3)) * 20).astype(np.float32)
    ctx.set_tuning("nn_variant", S)
    idx, d2 = icp.nearest_neighbours(src, tgt, ctx=ctx, culled=culled)
    ctx.set_tuning("nn_variant", 0)
    assert_nn_valid(src, tgt, idx, d2)


@pytest.mark.parametrize("n,m,shape", [(30000, 40000, "cube"), (20000, 60000, "sheet"), (50000, 50000, "clusters")])
def test_culled_equals_brute_force_bitwise(icp, ctx, n, m, shape):
    """The culled index must reproduce the brute-force kernel exactly (indices and distances), on clouds with very
    different spatial structure, and it must actually skip most tiles."""
    rng = np.random.default_rng(m)
    if shape == "cube":
        tgt = rng.random((m, 3)) * 20
        src = rng.random((n, 3)) * 22 - 1
    elif shape == "sheet":                                   # a thin slab: degenerate boxes
        tgt = np.stack([rng.random(m) * 30, rng.random(m) * 30, rng.normal(size=m) * 0.01], 1)
        src = np.stack([rng.random(n) * 30, rng.random(n) * 30, rng.normal(size=n) * 0.5], 1)
    else:
        centres = rng.normal(size=(12, 3)) * 15
        tgt = centres[rng.integers(0, 12, m)] + rng.normal(size=(m, 3)) * 0.3
        src = centres[rng.integers(0, 12, n)] + rng.normal(size=(n, 3)) * 0.6
    src, tgt = src.astype(np.float32), tgt.astype(np.float32)
    dev_b = icp.IcpDevice(src, tgt, ctx, culled=False)
    dev_b.nn()
    bi, bd = dev_b.download()
    dev_b.free()
    dev_c = icp.IcpDevice(src, tgt, ctx, culled=True)
    swept = dev_c.nn(want_stats=True)
    ci, cd = dev_c.download()
    dev_c.free()
    np.testing.assert_array_equal(ci, bi)
    np.testing.assert_array_equal(cd, bd)
    n_tiles = -(-m // 1024)
    n_groups = -(-n // 256)
    assert swept < 0.35 * n_tiles * n_groups, (swept, n_tiles * n_groups)


def test_index_query_unsorted_sources_and_permutation(icp, ctx):
    """The stateless path (presorted=0: the query sorts a copy itself) and the in-place sort + permutation."""
    rng = np.random.default_rng(31)
    tgt = (rng.random((30000, 3)) * 10).astype(np.float32)
    src = (rng.random((7000, 3)) * 10).astype(np.float32)
    want_i, want_d = OI.nearest_neighbours(src, tgt)
    d_tgt = ctx.alloc(tgt.nbytes).upload(tgt)
    d_src = ctx.alloc(src.nbytes).upload(src)
    d_idx, d_d2 = ctx.alloc(src.shape[0] * 4), ctx.alloc(src.shape[0] * 4)
    ix = icp.NNIndex(ctx, d_tgt.ptr, tgt.shape[0])
    ix.query(d_src.ptr, src.shape[0], d_idx.ptr, d_d2.ptr, presorted=False)
    np.testing.assert_array_equal(d_idx.download(np.uint32, src.shape[0]), want_i)
    np.testing.assert_allclose(d_d2.download(np.float32, src.shape[0]), want_d, rtol=2e-7)
    # a deliberately incoherent order still gives the right answer with presorted=1 (only slower)
    ix.query(d_src.ptr, src.shape[0], d_idx.ptr, d_d2.ptr, presorted=True)
    np.testing.assert_array_equal(d_idx.download(np.uint32, src.shape[0]), want_i)
    d_perm = ctx.alloc(src.shape[0] * 4)
    ix.sort_cloud(d_src.ptr, src.shape[0], d_perm.ptr)
    perm = d_perm.download(np.uint32, src.shape[0])
    assert sorted(perm.tolist()) == list(range(src.shape[0]))
    np.testing.assert_array_equal(d_src.download(np.float32, src.size).reshape(-1, 3), src[perm])
    swept = ix.query(d_src.ptr, src.shape[0], d_idx.ptr, d_d2.ptr, want_stats=True, presorted=True)
    np.testing.assert_array_equal(d_idx.download(np.uint32, src.shape[0]), want_i[perm])
    assert swept < 0.6 * (-(-src.shape[0] // 256)) * (-(-tgt.shape[0] // 1024))   # 30 tiles only: culling is modest here
    ix.close()
    for b in (d_tgt, d_src, d_idx, d_d2, d_perm):
        b.free()


def test_culled_cross_tile_ties_and_duplicates(icp, ctx):
    rng = np.random.default_rng(9)
    base = rng.integers(0, 6, (300, 3)).astype(np.float32)      # heavy duplication on an integer lattice
    tgt = np.tile(base, (20, 1))                                 # every point at 20 original indices, many tiles
    src = np.concatenate([base + np.float32(0.25), base])        # equidistant neighbours + exact hits
    idx_b, d2_b = icp.nearest_neighbours(src, tgt, ctx=ctx, culled=False)
    idx_c, d2_c = icp.nearest_neighbours(src, tgt, ctx=ctx, culled=True)
    np.testing.assert_array_equal(idx_c, idx_b)
    np.testing.assert_array_equal(d2_c, d2_b)
    np.testing.assert_array_equal(idx_c, OI.nearest_neighbours(src, tgt)[0])
    assert idx_c.max() < 300


def test_nn_ties_pick_lowest_index(icp, ctx):
    rng = np.random.default_rng(3)
    base = (rng.integers(0, 8, (40, 3))).astype(np.float32)       # small integer lattice: exact arithmetic
    tgt = np.concatenate([base, base, base[::-1]])                  # every point appears at >= 3 indices
    src = base + np.float32(0.25)
    idx, d2 = icp.nearest_neighbours(src, tgt, ctx=ctx)
    oi, od = OI.nearest_neighbours(src, tgt)
    np.testing.assert_array_equal(idx, oi)
    np.testing.assert_array_equal(d2, od)
    # duplicates spread over different 32-target groups and 1024-target tiles
    tgt2 = np.tile(base, (60, 1))
    idx2, _ = icp.nearest_neighbours(src, tgt2, ctx=ctx)
    np.testing.assert_array_equal(idx2, OI.nearest_neighbours(src, tgt2)[0])
    assert idx2.max() < 40


def test_nn_empty_source_and_errors(R, icp, ctx):
    idx, d2 = icp.nearest_neighbours(np.zeros((0, 3), np.float32), np.zeros((4, 3), np.float32), ctx=ctx)
    assert idx.shape == (0,) and d2.shape == (0,)
    with pytest.raises(R.R3DError):
        icp.nearest_neighbours(np.zeros((2, 3), np.float32), np.zeros((0, 3), np.float32), ctx=ctx)


def test_accumulate_matches_oracle_and_is_deterministic(icp, ctx):
    src, tgt, _, _ = OI.synthetic_pair(n_tgt=20000, n_src=15000, noise=0.01)
    dev = icp.IcpDevice(src, tgt, ctx)
    dev.nn()
    idx, d2 = dev.download()
    s1 = dev.sums()
    s2 = dev.sums()
    np.testing.assert_array_equal(s1, s2)                       # fixed reduction tree: bitwise repeatable
    want = OI.pair_sums(src, tgt, idx)
    np.testing.assert_allclose(s1, want, rtol=1e-12, atol=1e-9)
    gate = float(np.median(d2))
    sg = dev.sums(gate)
    np.testing.assert_allclose(sg, OI.pair_sums(src, tgt, idx, d2, gate), rtol=1e-12, atol=1e-9)
    assert 0 < sg[0] < s1[0]
    dev.free()


def test_known_correspondences_give_the_closed_form_answer(icp, ctx):
    """src is an exact similarity image (s=1.7, 10 deg, |t|=0.5: SURVEY C3) of a subset of tgt.  With the
    TRUE correspondences the 18 sums from the GPU must give (s, R, t) back to fp32 data precision."""
    src, tgt, T_true, pick = OI.synthetic_pair(n_tgt=6000, n_src=4000, s=1.7, angle_deg=10.0, t_norm=0.5)
    dev = icp.IcpDevice(src, tgt, ctx, culled=False)      # keep the source order: d_idx is uploaded by hand
    dev.d_idx.upload(pick.astype(np.uint32))
    sums = dev.sums()
    dev.free()
    np.testing.assert_allclose(sums, OI.pair_sums(src, tgt, pick.astype(np.uint32)), rtol=1e-12, atol=1e-9)
    T = icp.umeyama_from_sums(sums)
    np.testing.assert_allclose(T, T_true, rtol=0, atol=5e-5)


def test_icp_loop_converges_and_matches_oracle_loop(icp, ctx):
    """A small misalignment (inside ICP's basin: displacement < half the point spacing): the GPU loop must
    land on the true transform and agree with the oracle's loop step for step."""
    src, tgt, T_true, _ = OI.synthetic_pair(n_tgt=3000, n_src=2500, s=1.01, angle_deg=0.5, t_norm=0.02, seed=4)
    T, info = icp.icp_similarity(src, tgt, max_iter=40, ctx=ctx)
    np.testing.assert_allclose(T, T_true, rtol=0, atol=2e-4)
    assert info["rms_history"][-1] < 1e-4 and info["rms_history"][0] > 10 * info["rms_history"][-1]
    np.testing.assert_allclose(T, OI.icp_similarity(src, tgt, max_iter=40), rtol=0, atol=5e-5)


def test_icp_with_noise_matches_oracle_loop(icp, ctx):
    src, tgt, T_true, _ = OI.synthetic_pair(n_tgt=5000, n_src=3000, s=1.01, angle_deg=0.5, t_norm=0.02, noise=0.01,
                                            seed=21)
    T, info = icp.icp_similarity(src, tgt, max_iter=25, ctx=ctx)
    T_ref = OI.icp_similarity(src, tgt, max_iter=25)
    np.testing.assert_allclose(T, T_ref, rtol=0, atol=5e-5)
    assert np.abs(T - T_true).max() < 5e-3


def test_objects_outliving_their_context_do_not_crash(R, icp):
    """Handles tied to a context (NN index, voxel set, device buffers) may be garbage collected after it: no use-after-free."""
    import importlib
    V = importlib.import_module(R.__name__ + ".voxelmap")
    c = R.Context(0)
    tgt = np.random.default_rng(0).random((3000, 3)).astype(np.float32)
    buf = c.alloc(tgt.nbytes).upload(tgt)
    ix = icp.NNIndex(c, buf.ptr, 3000)
    vs = V.VoxelSet(0.1, 4096, c)
    vs.insert(tgt)
    c.close()                       # closes ix and vs first
    assert ix.handle is None and vs.handle is None
    del ix, vs, buf                 # finalizers run against a closed context
