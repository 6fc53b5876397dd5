"""ICP similarity estimation on the MI355X: produces the T_data.txt that the reference's
other_tools/transfer_T_icp.py consumes (get_T, icp:33-43) but never computes itself -- the
reference obtained it from an external tool (readme: CloudCompare).  Build-defined:

  repeat: nearest neighbour of every source point in the target cloud  (HIP, r3d_icp_nn)
          18 fp64 sums over the matched pairs                          (HIP, r3d_icp_accumulate)
          closed-form similarity (s, R, t) from the sums               (host, 3x3 SVD; Umeyama 1991)
          move the source cloud by it                                  (HIP, r3d_apply_T, in place)

Both clouds stay resident in HBM for the whole loop; per iteration only 144 bytes come back.
"""
import ctypes as C

import numpy as np

from . import _lib as L
from .device import default_context


def umeyama_from_sums(sums, with_scale=True):
    """4x4 T = [sR t; 0 1] minimising sum w |q - (s R p + t)|^2 from the 18 sums of
    r3d_icp_accumulate: n, sum p (3), sum q (3), sum p_a q_b (9, a major), sum |p|^2, sum |q|^2.
    Solved by the library (r3d_umeyama_from_sums: fp64 one-sided Jacobi SVD) -- the same code the
    device-side solve of r3d_icp_iterate runs."""
    sums = np.ascontiguousarray(sums, dtype=np.float64)
    if sums.shape != (18,):
        raise ValueError("expected 18 sums")
    T = np.zeros(16, dtype=np.float64)
    rc = L.load().r3d_umeyama_from_sums(sums.ctypes.data, 1 if with_scale else 0, T.ctypes.data, None)
    if rc != L.OK:
        raise ValueError("need at least 3 matched pairs with some spread (weight sum %g)" % sums[0])
    return T.reshape(4, 4)


def swap_pair_sums(sums):
    """The 18 sums with the roles of p and q exchanged."""
    s = np.asarray(sums, dtype=np.float64)
    out = s.copy()
    out[1:4], out[4:7] = s[4:7], s[1:4]
    out[7:16] = s[7:16].reshape(3, 3).T.reshape(9)
    out[16], out[17] = s[17], s[16]
    return out


def moments_init(mom_src, mom_tgt):
    """Coarse similarity from per-cloud moments (sum p, sum |p|^2): centroid onto centroid, RMS radius onto RMS
    radius, no rotation.  mom_*: the 18 sums of a cloud paired with itself."""
    def stats(m):
        n = m[0]
        mu = m[1:4] / n
        return mu, np.sqrt(max(m[16] / n - mu @ mu, 0.0))
    mu_p, r_p = stats(np.asarray(mom_src, dtype=np.float64))
    mu_q, r_q = stats(np.asarray(mom_tgt, dtype=np.float64))
    s = r_q / r_p if r_p > 0 and r_q > 0 else 1.0
    T = np.eye(4)
    T[:3, :3] *= s
    T[:3, 3] = mu_q - s * mu_p
    return T


def _cloud_stats(m):
    """(centroid, RMS radius, covariance) of a cloud from its 18 self-paired sums."""
    m = np.asarray(m, dtype=np.float64)
    n = m[0]
    mu = m[1:4] / n
    cov = m[7:16].reshape(3, 3) / n - np.outer(mu, mu)
    return mu, np.sqrt(max(m[16] / n - mu @ mu, 0.0)), 0.5 * (cov + cov.T)


def init_candidates(mom_src, mom_tgt, with_scale=True):
    """Starting transforms for init="auto": [("moments", T0), 24 x ("pca" | "pca-permuted", T)].  T0 = moments_init (no
    rotation).  The others turn the source's principal axes onto the target's: the 24 proper rotations V_q S P V_p^T with S a
    sign matrix and P an axis permutation (eigenvectors are defined up to sign, mirror images are excluded; the permutations
    matter when two extents are alike -- a room seen only in part can swap its two longest axes).  That lets the estimator
    start from ANY relative orientation when the scene is anisotropic (a room, a street), e.g. a camera-frame cloud against
    COLMAP's arbitrary world gauge.  On a near-isotropic cloud the axes mean nothing; the cost test in icp_similarity then
    keeps T0."""
    mu_p, r_p, cov_p = _cloud_stats(mom_src)
    mu_q, r_q, cov_q = _cloud_stats(mom_tgt)
    s = (r_q / r_p if r_p > 0 and r_q > 0 else 1.0) if with_scale else 1.0

    def make(Rm):
        T = np.eye(4)
        T[:3, :3] = s * Rm
        T[:3, 3] = mu_q - s * (Rm @ mu_p)
        return T

    out = [("moments", make(np.eye(3)))]
    wp, Vp = np.linalg.eigh(cov_p)
    wq, Vq = np.linalg.eigh(cov_q)
    if not (np.all(np.isfinite(wp)) and np.all(np.isfinite(wq))) or wp[2] <= 0 or wq[2] <= 0:
        return out
    # proper rotations only: the third sign follows from the other two, the permutation's parity and the handedness of the
    # two eigenvector frames (no determinant per candidate: this runs inside the timed estimate)
    hand = (1.0 if np.linalg.det(Vq) > 0 else -1.0) * (1.0 if np.linalg.det(Vp) > 0 else -1.0)
    for perm, parity in (((0, 1, 2), 1.0), ((0, 2, 1), -1.0), ((1, 0, 2), -1.0), ((1, 2, 0), 1.0), ((2, 0, 1), 1.0), ((2, 1, 0), -1.0)):
        PVt = Vp.T[list(perm), :]                # P V_p^T: row a = source axis perm[a]
        for sx in (1.0, -1.0):
            for sy in (1.0, -1.0):
                sz = 1.0 if hand * parity * sx * sy > 0 else -1.0
                Rm = (Vq * np.array([sx, sy, sz])) @ PVt
                out.append(("pca" if perm == (0, 1, 2) else "pca-permuted", make(Rm)))
    return out


def trimmed_means_device(d_values_ptr, n_classes, per_class, keep, ctx):
    """Per consecutive block of per_class device floats: the mean of the finite values <= the block's `keep` order statistic
    ("lower" rule), selected and summed on the GPU (r3d_trimmed_means_f32); +inf for a block with no finite value."""
    out = np.empty(n_classes, dtype=np.float64)
    L.check(ctx.lib.r3d_trimmed_means_f32(ctx.handle, d_values_ptr, int(n_classes), int(per_class), float(keep), out.ctypes.data))
    return out


INIT_SAMPLES, INIT_KEEP, INIT_PREFER_MOMENTS = 8192, 0.8, 1.05


class NNIndex:
    """Spatially culled exact nearest-neighbour index over a device-resident target cloud (r3d_nn_index)."""

    def __init__(self, ctx, d_tgt_ptr, n_tgt):
        self.ctx = ctx
        h = C.c_void_p()
        L.check(ctx.lib.r3d_nn_index_create(ctx.handle, d_tgt_ptr, int(n_tgt), C.byref(h)))
        self.handle = h.value
        ctx.adopt(self)

    def query(self, d_src_ptr, n_src, d_idx_ptr, d_d2_ptr, want_stats=False, presorted=False):
        swept = C.c_int64()
        L.check(self.ctx.lib.r3d_nn_index_query(self.handle, d_src_ptr, int(n_src), d_idx_ptr, d_d2_ptr,
                                                1 if presorted else 0, C.byref(swept) if want_stats else None))
        return swept.value

    def rebuild(self, d_tgt_ptr, n_tgt):
        """Index another cloud of at most the size this one was created with, reusing the allocations."""
        L.check(self.ctx.lib.r3d_nn_index_rebuild(self.handle, d_tgt_ptr, int(n_tgt)))

    def sort_cloud(self, d_xyz_ptr, n, d_perm_ptr=None):
        L.check(self.ctx.lib.r3d_nn_index_sort_cloud(self.handle, d_xyz_ptr, int(n), d_perm_ptr))

    def sort_cloud_valid(self, d_xyz_ptr, n, d_perm_ptr=None):
        """sort_cloud for a cloud that still holds non-points: rows with a NaN / inf coordinate go behind the valid ones;
        returns how many valid rows are in front (synchronous)."""
        k = C.c_int64()
        L.check(self.ctx.lib.r3d_nn_index_sort_cloud_valid(self.handle, d_xyz_ptr, int(n), d_perm_ptr, C.byref(k)))
        return k.value

    def close(self):
        if self.handle:
            self.ctx.lib.r3d_nn_index_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


STATE_DOUBLES, STATE_HISTORY = 512, 48       # R3D_ICP_STATE_DOUBLES / R3D_ICP_STATE_HISTORY (include/r3d.h)


class _Slice:
    """A piece of an arena block: the DeviceBuffer surface (ptr, nbytes, upload, download) without an allocation of its own."""

    def __init__(self, ctx, ptr, nbytes):
        self.ctx, self.ptr, self.nbytes = ctx, ptr, int(nbytes)

    def upload(self, host):
        host = np.ascontiguousarray(host)
        assert host.nbytes <= self.nbytes
        L.check(self.ctx.lib.r3d_memcpy_h2d(self.ctx.handle, self.ptr, host.ctypes.data, host.nbytes))
        self.ctx.sync()  # `host` may be a temporary
        return self

    def download(self, dtype, count):
        out = np.empty(count, dtype=dtype)
        assert out.nbytes <= self.nbytes
        L.check(self.ctx.lib.r3d_memcpy_d2h(self.ctx.handle, out.ctypes.data, self.ptr, out.nbytes))
        self.ctx.sync()
        return out

    def free(self):
        pass


class _Arena:
    """Bump allocator over a few big HBM blocks: an estimate makes ~30 buffers, and every hipMalloc / hipFree pair (the free
    synchronises the device) is tens of microseconds -- a measurable share of a 10 ms job.  Everything goes at once."""

    def __init__(self, ctx, first_block):
        self.ctx, self.blocks, self.at, self.block_bytes = ctx, [], 0, max(int(first_block), 1 << 20)

    def take(self, nbytes):
        nbytes = (max(int(nbytes), 16) + 255) & ~255
        if not self.blocks or self.at + nbytes > self.blocks[-1].nbytes:
            self.blocks.append(self.ctx.alloc(max(nbytes, self.block_bytes)))
            self.at = 0
        s = _Slice(self.ctx, self.blocks[-1].ptr + self.at, nbytes)
        self.at += nbytes
        return s

    def free(self):
        for b in self.blocks:
            b.free()
        self.blocks = []


class IcpDevice:
    """Source / target clouds resident on one GPU.  culled=True (default) answers NN queries through the
    Morton-tile index (same results, far fewer pair evaluations); culled=False runs the plain brute-force sweep."""

    def __init__(self, src, tgt, ctx=None, culled=True):
        self.ctx = ctx or default_context()
        src = np.ascontiguousarray(src, dtype=np.float32)
        tgt = np.ascontiguousarray(tgt, dtype=np.float32)
        if src.ndim != 2 or src.shape[1] != 3 or tgt.ndim != 2 or tgt.shape[1] != 3:
            raise ValueError("clouds must be [N,3]")
        if tgt.shape[0] < 1:
            raise ValueError("target cloud is empty")
        self.n, self.m = src.shape[0], tgt.shape[0]
        self._tgt_host = tgt             # the target never moves: target_spacing() samples it without a D2H
        self._src_host = src             # the source as handed over (original order and frame): init_costs() samples it
        c = self.ctx
        # one block for the clouds, the match arrays and (culled estimates) the work buffers of every stage
        self._arena = _Arena(c, 24 * self.n + (66 if culled else 14) * self.m + (12 << 20 if culled else 1 << 16))
        take = self._arena.take
        self.d_src = take(max(src.nbytes, 16)).upload(src)
        self.d_tgt = take(tgt.nbytes).upload(tgt)
        self.d_idx = take(max(self.n * 4, 16))
        self.d_d2 = take(max(self.n * 4, 16))
        self.d_sums = take(18 * 8)
        self.d_state = take(STATE_DOUBLES * 8)
        self.index = NNIndex(c, self.d_tgt.ptr, self.m) if culled else None
        self.d_perm = None
        self._back = None            # lazily built: reverse-direction search (target -> source snapshot)
        self._pool = {}              # named grow-only device buffers: no hipMalloc / hipFree (each a sync) inside an estimate
        self._src_index = None       # index of the source AS UPLOADED (sorted order), shared by init_costs and the back search
        self._spacing = None
        self._spacing_pending, self._probe_index = False, None
        self.trace = None            # a list: _tick(name) then syncs and appends (name, seconds) -- icp_similarity(profile=True)
        self._src_index_rows = None  # "upload": the source index was built before sort_source() rearranged d_src; "sorted": after
        self.d_inv = None            # inverse of d_perm (upload row -> current row), made when the back search needs it
        self._take = take

    def sort_source(self):
        """Put the source cloud, WHERE IT LIES NOW, into the target index's Morton order -- once: rigid / similarity moves keep
        every workgroup's 256 sources a compact blob, so no later query needs to sort.  d_perm maps back to the upload order.
        The estimator calls this after its initial move: sorted where it was uploaded (another unit, another place: most
        coordinates clamp to the target's quantisation frame) the cloud is NOT coherent, and every later query swept twice the
        tiles (0.73 instead of 0.33 ms per iteration at 500k, found in round 3).  nn() / iterate() call it if nobody has."""
        if self.index is None or not self.n or self.d_perm is not None:
            return
        self.d_perm = self._take(self.n * 4)
        self.index.sort_cloud(self.d_src.ptr, self.n, self.d_perm.ptr)

    def _tick(self, name):
        if self.trace is not None:
            import time
            self.ctx.sync()
            self.trace.append((name, time.perf_counter()))

    def _buf(self, name, nbytes):
        """Named work buffer from the arena (kept for the life of this object; asked again with a larger size it is re-taken)."""
        b = self._pool.get(name)
        if b is None or b.nbytes < nbytes:
            b = self._arena.take(nbytes)
            self._pool[name] = b
        return b

    def source_index(self):
        """Index of the source cloud as it was uploaded (call before the first move, or not at all)."""
        if self._src_index is None:
            self._src_index = NNIndex(self.ctx, self.d_src.ptr, self.n)
            self._src_index_rows = "upload" if self.d_perm is None else "sorted"
        return self._src_index

    def nn(self, want_stats=False):
        c = self.ctx
        self.sort_source()
        if self.index is not None:
            return self.index.query(self.d_src.ptr, self.n, self.d_idx.ptr, self.d_d2.ptr, want_stats, presorted=True)
        L.check(c.lib.r3d_icp_nn(c.handle, self.d_src.ptr, self.n, self.d_tgt.ptr, self.m, self.d_idx.ptr,
                                 self.d_d2.ptr))
        return 0

    def _download_sums(self):
        return self.d_sums.download(np.float64, 18)

    def sums(self, max_d2=-1.0, dead_zone=0.0):
        """The 18 pair sums over the current matches (separate gather pass, r3d_icp_accumulate_dev)."""
        c = self.ctx
        need_d2 = max_d2 >= 0 or dead_zone > 0
        L.check(c.lib.r3d_icp_accumulate_dev(c.handle, self.d_src.ptr, self.n, self.d_tgt.ptr, self.m, self.d_idx.ptr,
                                             self.d_d2.ptr if need_d2 else None, float(max_d2), float(dead_zone),
                                             self.d_sums.ptr))
        return self._download_sums()

    def nn_sums(self, max_d2=-1.0, dead_zone=0.0):
        """Nearest neighbours AND the 18 sums in one pass (culled index: sums taken in the query kernel's epilogue)."""
        c = self.ctx
        self.sort_source()
        if self.index is not None:
            L.check(c.lib.r3d_nn_index_query_sums(self.index.handle, self.d_src.ptr, self.n, self.d_idx.ptr,
                                                  self.d_d2.ptr, 1, float(max_d2), float(dead_zone), self.d_sums.ptr))
            return self._download_sums()
        self.nn()
        return self.sums(max_d2, dead_zone)

    def moments(self, which):
        """The 18 sums of one cloud paired with itself: [0] n, [1:4] sum p, [7:16] sum p p^T, [16] sum |p|^2."""
        return self.moments_both()[0 if which == "src" else 1] if which == "both" else self._moments_one(which)

    def _moments_one(self, which):
        c = self.ctx
        buf, n = (self.d_src, self.n) if which == "src" else (self.d_tgt, self.m)
        L.check(c.lib.r3d_icp_accumulate_dev(c.handle, buf.ptr, n, buf.ptr, n, None, None, -1.0, 0.0, self.d_sums.ptr))
        return self._download_sums()

    def moments_both(self):
        """(source moments, target moments) with ONE trip to the host."""
        c = self.ctx
        d = self._buf("mom", 36 * 8)
        L.check(c.lib.r3d_icp_accumulate_dev(c.handle, self.d_src.ptr, self.n, self.d_src.ptr, self.n, None, None, -1.0, 0.0, d.ptr))
        L.check(c.lib.r3d_icp_accumulate_dev(c.handle, self.d_tgt.ptr, self.m, self.d_tgt.ptr, self.m, None, None, -1.0, 0.0,
                                             d.ptr + 18 * 8))
        both = d.download(np.float64, 36)
        return both[:18].copy(), both[18:].copy()

    def target_spacing_begin(self, max_queries=32768):
        """Enqueue the sampling-resolution probe of the target (see target_spacing) without waiting for it: strided row
        gathers, a probe index, one query and the selection of the median all run on the GPU while the host goes on (the
        estimator starts it first and reads it three stages later)."""
        if self._spacing is not None or self._spacing_pending or self.m < 8:
            return
        c = self.ctx
        n_base, n_half = self.m // 2, (self.m + 1) // 2           # tgt[1::2], tgt[0::2]
        step = max(1, n_half // max_queries)
        n_probe = -(-n_half // step)
        d_base, d_probe = self._buf("sp_base", n_base * 12), self._buf("sp_probe", n_probe * 12)
        d_i, d_d, d_o = self._buf("sp_idx", n_probe * 4), self._buf("sp_d2", n_probe * 4), self._buf("sp_out", 8)
        L.check(c.lib.r3d_gather_rows_strided(c.handle, self.d_tgt.ptr, self.m, 1, 2, n_base, d_base.ptr))
        L.check(c.lib.r3d_gather_rows_strided(c.handle, self.d_tgt.ptr, self.m, 0, 2 * step, n_probe, d_probe.ptr))
        self._probe_index = NNIndex(c, d_base.ptr, n_base)
        self._probe_index.query(d_probe.ptr, n_probe, d_i.ptr, d_d.ptr)
        L.check(c.lib.r3d_select_quantile_f32_dev(c.handle, d_d.ptr, n_probe, 0.5, d_o.ptr))
        self._spacing_pending = True

    def target_spacing(self, max_queries=32768):
        """"Lower" median nearest-neighbour distance between two interleaved halves of the target cloud: its sampling
        resolution, independent of how the clouds are aligned.  All on the GPU (strided row gathers, a probe index, the order
        statistic selected on the device; 8 bytes come back); cached: the target never moves."""
        if self._spacing is not None:
            return self._spacing
        if self.m < 8:
            self._spacing = 0.0
            return 0.0
        self.target_spacing_begin(max_queries)
        out = self._pool["sp_out"].download(np.uint32, 2)
        med, cnt = float(out[:1].view(np.float32)[0]), int(out[1])
        self._probe_index.close()
        self._probe_index, self._spacing_pending = None, False
        self._spacing = float(np.sqrt(med)) if cnt else 0.0
        return self._spacing

    def init_costs(self, transforms):
        """Symmetric, robust misfit of each candidate start (call BEFORE the source is moved): a strided sample of the
        source moved by T against the target index, plus a strided sample of the target moved by T^-1 against an index of
        the source (distances brought to target units); each side the mean of the lowest INIT_KEEP share of d2.
        All candidates share ONE query per direction (their moved samples are laid end to end): two queries, not ten."""
        c = self.ctx
        ss_host = self._src_host[::max(1, self.n // INIT_SAMPLES)]
        st_host = self._tgt_host[::max(1, self.m // INIT_SAMPLES)]
        k, ns, nt = len(transforms), ss_host.shape[0], st_host.shape[0]
        cap = k * max(ns, nt)
        d_ss, d_st = self._buf("ic_ss", ns * 12).upload(ss_host), self._buf("ic_st", nt * 12).upload(st_host)
        d_mv, d_i, d_d = self._buf("ic_mv", cap * 12), self._buf("ic_i", cap * 4), self._buf("ic_d", cap * 4)
        self._tick("ic.samples_up")
        ix_src = self.source_index()
        self._tick("ic.source_index")
        Ts = np.ascontiguousarray(np.stack([np.asarray(T, dtype=np.float64).reshape(4, 4) for T in transforms]))
        L.check(c.lib.r3d_apply_T_many(c.handle, d_ss.ptr, L.F32, ns, Ts.ctypes.data, k, d_mv.ptr, L.F32))
        self._tick("ic.fwd_moves")
        self.index.query(d_mv.ptr, k * ns, d_i.ptr, d_d.ptr)
        self._tick("ic.fwd_query")
        fwd = trimmed_means_device(d_d.ptr, k, ns, INIT_KEEP, c)
        self._tick("ic.fwd_means")
        invs = np.ascontiguousarray(np.linalg.inv(Ts))
        L.check(c.lib.r3d_apply_T_many(c.handle, d_st.ptr, L.F32, nt, invs.ctypes.data, k, d_mv.ptr, L.F32))
        self._tick("ic.back_moves")
        ix_src.query(d_mv.ptr, k * nt, d_i.ptr, d_d.ptr)
        self._tick("ic.back_query")
        back = trimmed_means_device(d_d.ptr, k, nt, INIT_KEEP, c)
        self._tick("ic.back_means")
        s2 = np.cbrt(np.abs(np.linalg.det(Ts[:, :3, :3]))) ** 2
        return (fwd + back * s2).tolist()

    # ---- reverse direction (symmetric coarse phase): every TARGET point's nearest point of a source snapshot ----
    def back_begin(self, T_since=None):
        """Reverse direction: every TARGET point's nearest point of the source.  The source is searched through the index of
        its UPLOADED state (source_index(): built once, shared with init_costs); back_sums takes the target into that frame
        by the inverse of everything the source has been moved by since (T_since).  Keeps a copy of the target whose ROW ORDER
        is the index's Morton order of the target as it lies in that frame now (a workgroup's 256 queries stay a compact blob
        under the small moves that follow -- sorting the world-frame target by a frame it does not lie in cost 4x in culling)."""
        c = self.ctx
        ix = self.source_index()
        d_tgt_b, d_moved = self._buf("bk_tgt", self.m * 12), self._buf("bk_moved", self.m * 12)
        d_perm = self._buf("bk_perm", self.m * 4)
        T_inv = np.ascontiguousarray(np.linalg.inv(np.eye(4) if T_since is None else T_since), dtype=np.float64)
        L.check(c.lib.r3d_apply_T(c.handle, self.d_tgt.ptr, L.F32, self.m, T_inv.ctypes.data, d_moved.ptr, L.F32))
        ix.sort_cloud(d_moved.ptr, self.m, d_perm.ptr)
        L.check(c.lib.r3d_gather_rows(c.handle, self.d_tgt.ptr, self.m, d_perm.ptr, self.m, d_tgt_b.ptr))
        self.sort_source()
        if self._src_index_rows == "upload" and self.d_perm is not None and self.d_inv is None:
            # the index reports rows of the cloud as uploaded; d_src has been rearranged since: upload row -> current row
            self.d_inv = self._take(self.n * 4)
            L.check(c.lib.r3d_permutation_invert(c.handle, self.d_perm.ptr, self.n, self.d_inv.ptr))
        self._back = {"index": ix, "tgt": d_tgt_b, "moved": d_moved, "idx": self._buf("bk_idx", self.m * 4),
                      "d2": self._buf("bk_d2", self.m * 4), "sums": self._buf("bk_sums", 18 * 8)}

    def d2_quantile(self, q):
        """The q order statistic ("lower" rule) of the current source -> target squared match distances (after nn() /
        nn_sums()), selected on the GPU; -1 when no distance is finite."""
        v, cnt = select_quantile(self.d_d2.ptr, self.n, q, self.ctx)
        return float(v) if cnt else -1.0

    def back_sums(self, T_since, dead_zone=0.0, trim=None):
        """18 sums over the pairs (p = CURRENT source point nearest to target point q, q): the target is taken
        into the source index's frame by T_since^-1 (T_since = everything the source was moved by since it was uploaded),
        searched there, and the sums are formed in the world frame.
        trim: keep only the pairs up to that order statistic of the match distances (outlier rejection)."""
        c, b = self.ctx, self._back
        T_inv = np.ascontiguousarray(np.linalg.inv(T_since), dtype=np.float64)
        s_since = float(np.cbrt(abs(np.linalg.det(T_since[:3, :3]))))
        L.check(c.lib.r3d_apply_T(c.handle, b["tgt"].ptr, L.F32, self.m, T_inv.ctypes.data, b["moved"].ptr, L.F32))
        self._tick("back.move")
        b["index"].query(b["moved"].ptr, self.m, b["idx"].ptr, b["d2"].ptr, presorted=True)
        if self.d_inv is not None and self._src_index_rows == "upload":
            L.check(c.lib.r3d_remap_u32(c.handle, b["idx"].ptr, self.m, self.d_inv.ptr, self.n))
        self._tick("back.query")
        gate = -1.0
        if trim is not None and trim < 1.0:
            v, cnt = select_quantile(b["d2"].ptr, self.m, trim, c)
            gate = float(v) if cnt else -1.0
        # roles swapped: "src" = target rows (world frame), "tgt" = current source rows; d2 lives in the index's frame
        L.check(c.lib.r3d_icp_accumulate_dev(c.handle, b["tgt"].ptr, self.m, self.d_src.ptr, self.n, b["idx"].ptr,
                                             b["d2"].ptr, gate, float(dead_zone) / s_since if dead_zone > 0 else 0.0,
                                             b["sums"].ptr))
        return swap_pair_sums(b["sums"].download(np.float64, 18))

    def back_end(self):
        self._back = None          # its buffers live in the pool, its index is source_index()

    # ---- whole iterations on the GPU, no host round trip ----
    def state_reset(self):
        c = self.ctx
        L.check(c.lib.r3d_icp_state_reset(c.handle, self.d_state.ptr))

    def iterate(self, n_iters, with_scale=True, max_d2=-1.0):
        c = self.ctx
        self.sort_source()
        L.check(c.lib.r3d_icp_iterate(c.handle, self.index.handle if self.index is not None else None, self.d_src.ptr,
                                      self.n, self.d_tgt.ptr, self.m, self.d_idx.ptr, self.d_d2.ptr, int(n_iters),
                                      1 if with_scale else 0, float(max_d2), self.d_state.ptr))

    def state(self):
        st = self.d_state.download(np.float64, STATE_DOUBLES)
        it = int(st[32])
        return {"T_total": st[0:16].reshape(4, 4).copy(), "T_step": st[16:32].reshape(4, 4).copy(), "iterations": it,
                "degenerate": bool(st[33]), "rms": float(st[34]), "pairs": float(st[35]),
                "rms_history": st[STATE_HISTORY:STATE_HISTORY + min(it, STATE_DOUBLES - STATE_HISTORY)].tolist()}

    def move_source(self, T):
        c = self.ctx
        T = np.ascontiguousarray(T, dtype=np.float64)
        L.check(c.lib.r3d_apply_T(c.handle, self.d_src.ptr, L.F32, self.n, T.ctypes.data, self.d_src.ptr, L.F32))

    def _unpermute(self, a):
        if self.d_perm is None:
            return a
        perm = self.d_perm.download(np.uint32, self.n)
        out = np.empty_like(a)
        out[perm] = a
        return out

    def download(self):
        """(idx, d2) in the ORIGINAL source order."""
        idx = self.d_idx.download(np.uint32, self.n)
        d2 = self.d_d2.download(np.float32, self.n)
        return self._unpermute(idx), self._unpermute(d2)

    def source(self):
        """Current source cloud in the original order."""
        return self._unpermute(self.d_src.download(np.float32, self.n * 3).reshape(-1, 3))

    def free(self):
        self.back_end()
        if self.index is not None:
            self.index.close()
        if self._src_index is not None:
            self._src_index.close()
            self._src_index = None
        if self._probe_index is not None:
            self._probe_index.close()
            self._probe_index = None
        self._pool = {}
        self.ctx.sync()
        self._arena.free()


def nearest_neighbours(src, tgt, ctx=None, culled=False):
    """(idx uint32 [N], d2 float32 [N]): squared-L2 nearest target of every source point, lowest index on ties.
    culled=False: plain brute-force sweep (r3d_icp_nn_host); culled=True: Morton-tile index, same answer."""
    ctx = ctx or default_context()
    src = np.ascontiguousarray(src, dtype=np.float32)
    tgt = np.ascontiguousarray(tgt, dtype=np.float32)
    idx = np.empty(src.shape[0], dtype=np.uint32)
    d2 = np.empty(src.shape[0], dtype=np.float32)
    if src.shape[0] == 0:
        return idx, d2
    if culled:
        dev = IcpDevice(src, tgt, ctx, culled=True)
        try:
            dev.nn()
            return dev.download()
        finally:
            dev.free()
    L.check(ctx.lib.r3d_icp_nn_host(ctx.handle, src.ctypes.data, src.shape[0], tgt.ctypes.data, tgt.shape[0],
                                    idx.ctypes.data, d2.ctypes.data))
    return idx, d2


def _step_size(T, extent):
    """How far a 4x4 step is from the identity: linear part absolutely, translation relative to the cloud size."""
    return max(float(np.abs(T[:3, :3] - np.eye(3)).max()), float(np.abs(T[:3, 3]).max()) / max(extent, 1e-30))


def icp_similarity(src, tgt, max_iter=60, tol=1e-7, with_scale=True, trim_d2=None, ctx=None, culled=True, init="auto",
                   check_every=4, dead_zone=None, max_coarse=40, coarse_tol=2e-4, trim=None, coarse_trim=None,
                   profile=False):
    """Similarity (s, R, t) that maps `src` onto `tgt`: the T_data.txt of transfer_T_icp.py:99-108.
    Returns (T 4x4, info dict).  Three stages, all on device-resident clouds:

      init   "identity": none.  "moments": centroid onto centroid and RMS radius onto RMS radius (closes a scale gap
             such as monocular depth vs COLMAP units).  "auto" (default): the better of the moments start and the four
             principal-axis alignments (init_candidates: any relative orientation on an anisotropic scene), judged by a
             symmetric trimmed misfit on samples, then the symmetric dead-zone stage.
             A 4x4 array: that transform.
      coarse (init="auto") SYMMETRIC ICP under the cost max(0, d - dead_zone)^2: source -> target and target ->
             source matches together, each weighted w = max(0, 1 - dead_zone/d).  Matches closer than the target's
             sampling resolution (dead_zone, default 2 x target_spacing()) carry no weight, so the extents of the two
             clouds decide -- plain ICP stalls on a densely, evenly sampled volume, and one-directional ICP with a free
             scale can always shrink the source into the target.  Host solve per step (a handful of steps).  Needs the
             culled index (culled=True); with the brute-force search "auto" degrades to "moments".
      fine   plain ICP, nearest neighbour -> 18 sums -> Umeyama -> move, until the RMS match distance stops improving
             by more than `tol` (relative).  Runs in blocks of `check_every` iterations enqueued with no host round trip
             (r3d_icp_iterate: fused NN + sums, device-side solve); the host only reads the rms history per block.
    trim_d2: ignore pairs whose squared distance exceeds it in the fine stage (None = use all).
    trim: outlier rejection by rank -- in every step of both stages only the pairs up to that quantile of the current
          match distances take part (e.g. 0.9 drops the worst 10 %; least squares has no defence against gross outliers,
          and the coarse stage by construction listens to the FAR pairs).  None = use all.  The coarse stage trims more
          gently, at coarse_trim (default 1 - (1 - trim)/4): there the far pairs ARE the signal.  Reliable on structured
          scenes (surfaces); on a featureless uniform volume outliers and extent mismatch look alike and coarse_trim has
          to match the outlier share.
    profile: info["timings_ms"] = wall time per stage (upload + index build, init, spacing probe, coarse, fine), with a
          device sync at every stage boundary (off by default: the syncs cost a little)."""
    import time
    marks = [("start", time.perf_counter())]

    def mark(name):
        if profile:
            dev.ctx.sync()
            marks.append((name, time.perf_counter()))
            dev.trace.append((name, marks[-1][1]))

    dev = IcpDevice(src, tgt, ctx, culled)
    if profile:
        dev.trace = [("start", marks[0][1])]
    info = {"init": init if isinstance(init, str) else "matrix", "coarse_iterations": 0, "coarse_history": []}
    T_total = np.eye(4)
    try:
        mark("upload_index")
        if dev.n < 3:
            raise ValueError("need at least 3 source points")
        mode = init if isinstance(init, str) else "matrix"
        if mode not in ("identity", "moments", "auto", "matrix"):
            raise ValueError("init must be 'identity', 'moments', 'auto' or a 4x4 matrix")
        mom_s = mom_t = None
        if mode == "auto" and dev.index is not None:
            dev.source_index()               # of the source as uploaded: must exist before the first move
            if dead_zone is None:
                dev.target_spacing_begin()   # runs on the GPU while the host prepares the multi-start
        if mode in ("moments", "auto"):
            mom_s, mom_t = dev.moments_both()
        if mode == "matrix":
            T_total = np.array(init, dtype=np.float64).reshape(4, 4)
            dev.move_source(T_total)
        elif mode in ("moments", "auto"):
            dev._tick("init.moments")
            T_total = moments_init(mom_s, mom_t)
            if not with_scale:
                T_total = np.eye(4)
                T_total[:3, 3] = mom_t[1:4] / mom_t[0] - mom_s[1:4] / mom_s[0]
            if mode == "auto" and dev.index is not None and min(dev.n, dev.m) >= 16:
                # multi-start: the moments transform and the four principal-axis alignments, judged by a symmetric
                # trimmed misfit on samples; the plain moments start keeps the job unless an axis alignment is clearly better
                cands = init_candidates(mom_s, mom_t, with_scale)
                dev._tick("init.candidates")
                costs = dev.init_costs([T for _n, T in cands])
                best = int(np.argmin(costs))
                if costs[0] <= INIT_PREFER_MOMENTS * costs[best]:
                    best = 0
                info["init_candidates"] = [(cands[k][0], float(costs[k])) for k in range(len(cands))]
                info["init_choice"] = best
                T_total = cands[best][1]
            dev.move_source(T_total)
        dev.sort_source()                    # where the source lies NOW (after the initial move), once
        mark("init")
        if mode == "auto" and dev.index is not None:
            mu = mom_t[1:4] / mom_t[0]
            extent = float(np.sqrt(max(mom_t[16] / mom_t[0] - mu @ mu, 0.0)))
            d0 = float(dead_zone) if dead_zone is not None else 2.0 * dev.target_spacing()
            info["dead_zone"] = d0
            mark("spacing_probe")
            if d0 > 0 and extent > 0:
                T_since = T_total.copy()      # the back search runs in the frame of the source as uploaded
                dev.back_begin(T_since)
                dev._tick("coarse.back_begin")
                ct = coarse_trim if coarse_trim is not None else (None if trim is None else 1.0 - (1.0 - trim) / 4.0)
                info["coarse_trim"] = ct
                for _ in range(max_coarse):
                    if ct is not None and ct < 1.0:
                        dev.nn()
                        sums = dev.sums(dev.d2_quantile(ct), d0) + dev.back_sums(T_since, d0, ct)
                    else:
                        sums = dev.nn_sums(-1.0, d0)
                        dev._tick("coarse.fwd")
                        sums = sums + dev.back_sums(T_since, d0)
                        dev._tick("coarse.back")
                    info["coarse_history"].append(float(sums[0]))
                    if not sums[0] >= 3.0:
                        break                       # every match is inside the dead zone: extents agree
                    try:
                        T = umeyama_from_sums(sums, with_scale)
                    except ValueError:
                        break
                    dev.move_source(T)
                    T_total = T @ T_total
                    T_since = T @ T_since
                    info["coarse_iterations"] += 1
                    if _step_size(T, extent) <= coarse_tol:
                        break
                dev.back_end()
            mark("coarse")
        # fine stage
        dev.state_reset()
        dev._tick("fine.reset")
        max_d2 = -1.0 if trim_d2 is None else float(trim_d2)
        done, stop_at = 0, None
        while done < max_iter and stop_at is None:
            k = min(max(int(check_every), 1), max_iter - done)
            if trim is not None and trim < 1.0:          # re-rank once per block
                dev.nn()
                max_d2 = dev.d2_quantile(trim)
                if trim_d2 is not None:
                    max_d2 = min(max_d2, float(trim_d2))
            dev.iterate(k, with_scale, max_d2)
            dev._tick("fine.iterate")
            done += k
            st = dev.state()
            dev._tick("fine.state")
            h = st["rms_history"]
            for i in range(max(1, done - k), len(h)):
                if abs(h[i - 1] - h[i]) <= tol * max(h[i - 1], 1e-30):
                    stop_at = i
                    break
        st = dev.state()
        if st["degenerate"]:
            raise ValueError("ICP step undefined: fewer than 3 matched pairs (or no spread) after gating")
        T_total = st["T_total"] @ T_total
        mark("fine")
        if profile:
            info["timings_ms"] = {b[0]: (b[1] - a[1]) * 1e3 for a, b in zip(marks, marks[1:])}
            info["trace_ms"] = [(b[0], round((b[1] - a[1]) * 1e3, 3)) for a, b in zip(dev.trace, dev.trace[1:])]
        info.update({"iterations": st["iterations"], "rms_history": st["rms_history"], "degenerate": st["degenerate"],
                     "converged_at": stop_at})
    finally:
        dev.free()
    return T_total, info


# ---- rigid point-to-plane registration of two partially overlapping single views (readme.md:25; icp:99-108) ----
PLANE_SUMS = 29


def plane_step_from_sums(sums):
    """4x4 rigid step from the 29 sums of r3d_icp_plane_accumulate (library host code: the same arithmetic the device-side
    solve runs).  Raises ValueError when the matched normals leave a freedom unconstrained."""
    sums = np.ascontiguousarray(sums, dtype=np.float64)
    if sums.shape != (PLANE_SUMS,):
        raise ValueError("expected 29 sums")
    T = np.zeros(16, dtype=np.float64)
    rms = C.c_double()
    rc = L.load().r3d_plane_step_from_sums(sums.ctypes.data, T.ctypes.data, C.byref(rms))
    if rc != L.OK:
        raise ValueError(L.last_error())
    return T.reshape(4, 4), rms.value


def organized_normals(xyz, height, width, max_jump=0.05, viewpoint=None, ctx=None):
    """Normals [F*H*W,3] float32 of an organised cloud (raster order of gentxtcord, p2c:34-44): r3d_normals_organized."""
    ctx = ctx or default_context()
    xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
    n = xyz.shape[0]
    if n % (height * width) != 0:
        raise ValueError("cloud of %d points is not a whole number of %dx%d rasters" % (n, height, width))
    out = np.zeros((n, 3), dtype=np.float32)
    if n == 0:
        return out
    d_in, d_out = ctx.alloc(xyz.nbytes).upload(xyz), ctx.alloc(xyz.nbytes)
    try:
        vp = None if viewpoint is None else np.ascontiguousarray(viewpoint, dtype=np.float64)
        L.check(ctx.lib.r3d_normals_organized(ctx.handle, d_in.ptr, n // (height * width), int(height), int(width),
                                              float(max_jump), None if vp is None else vp.ctypes.data, d_out.ptr))
        out = d_out.download(np.float32, n * 3).reshape(-1, 3)
    finally:
        d_in.free()
        d_out.free()
    return out


def select_quantile(d_values_ptr, n, q, ctx=None):
    """(value, count): the element of rank floor(q (m - 1)) of the m finite values of a DEVICE float32 array, selected on the
    GPU (four passes of 8-bit digits; 8 bytes come back)."""
    ctx = ctx or default_context()
    v, m = C.c_float(), C.c_int64()
    L.check(ctx.lib.r3d_select_quantile_f32(ctx.handle, d_values_ptr, int(n), float(q), C.byref(v), C.byref(m)))
    return v.value, m.value


class PlaneIcpDevice:
    """Source cloud + organised target cloud (with its normals) resident on one GPU for point-to-plane ICP."""

    def __init__(self, src, tgt, tgt_shape=None, tgt_normals=None, max_jump=0.05, ctx=None, init=None, drop_invalid=False):
        """init: 4x4 applied to the source BEFORE it is put into the index's Morton order (the order has to be taken where the
        cloud lies when the queries run; rigid moves afterwards preserve it).
        drop_invalid: the source may hold rows that are no points -- (0,0,0) rows (pixels without depth, which gentxtcord emits
        like any other) and rows with a NaN / inf coordinate: they are taken out ON THE GPU (marked before the move, sorted
        behind the valid rows, self.n = the valid count) instead of by a host pass over the cloud."""
        self.ctx = c = ctx or default_context()
        src = np.ascontiguousarray(src, dtype=np.float32).reshape(-1, 3)
        tgt = np.ascontiguousarray(tgt, dtype=np.float32).reshape(-1, 3)
        if tgt.shape[0] < 1 or src.shape[0] < 6:
            raise ValueError("need a target cloud and at least 6 source points")
        self.n, self.m = src.shape[0], tgt.shape[0]
        self._arena = _Arena(c, 36 * self.n + 24 * self.m + (1 << 16))      # one block for everything below
        take = self._arena.take
        self.d_src = take(src.nbytes).upload(src)            # moved cloud, index order
        self.d_src0 = take(src.nbytes)                       # the cloud as it was at the last state reset
        self.d_tgt = take(tgt.nbytes).upload(tgt)
        self.d_nrm = take(tgt.nbytes)
        if tgt_normals is not None:
            nrm = np.ascontiguousarray(tgt_normals, dtype=np.float32).reshape(-1, 3)
            if nrm.shape != tgt.shape:
                raise ValueError("one normal per target point")
            self.d_nrm.upload(nrm)
        else:
            if tgt_shape is None or int(tgt_shape[0]) * int(tgt_shape[1]) != self.m:
                raise ValueError("an organised target needs tgt_shape=(H, W) with H*W == number of target points "
                                 "(or pass tgt_normals)")
            L.check(c.lib.r3d_normals_organized(c.handle, self.d_tgt.ptr, 1, int(tgt_shape[0]), int(tgt_shape[1]),
                                                float(max_jump), None, self.d_nrm.ptr))
        self.d_idx, self.d_d2 = take(self.n * 4), take(self.n * 4)
        self.d_sums = take(PLANE_SUMS * 8)
        self.d_state = take(STATE_DOUBLES * 8)
        self.index = NNIndex(c, self.d_tgt.ptr, self.m)
        self.d_perm = take(self.n * 4)
        if drop_invalid:
            L.check(c.lib.r3d_cloud_zero_rows_to_nan(c.handle, self.d_src.ptr, self.n))
        if init is not None:
            self.move_source(init)
        if drop_invalid:
            self.n_rows = self.n
            self.n = self.index.sort_cloud_valid(self.d_src.ptr, self.n, self.d_perm.ptr)
            if self.n < 6:
                self.free()
                raise ValueError("fewer than 6 source rows are points")
        else:
            self.index.sort_cloud(self.d_src.ptr, self.n, self.d_perm.ptr)

    def move_source(self, T):
        c = self.ctx
        T = np.ascontiguousarray(T, dtype=np.float64)
        L.check(c.lib.r3d_apply_T(c.handle, self.d_src.ptr, L.F32, self.n, T.ctypes.data, self.d_src.ptr, L.F32))

    def state_reset(self):
        c = self.ctx
        L.check(c.lib.r3d_icp_state_reset(c.handle, self.d_state.ptr))
        L.check(c.lib.r3d_memcpy_d2d(c.handle, self.d_src0.ptr, self.d_src.ptr, self.n * 12))

    def nn(self):
        self.index.query(self.d_src.ptr, self.n, self.d_idx.ptr, self.d_d2.ptr, presorted=True)

    def sums(self, trim_q=0.5, gate_scale=20.0, max_d2=-1.0):
        c = self.ctx
        L.check(c.lib.r3d_icp_plane_accumulate(c.handle, self.d_src.ptr, self.n, self.d_tgt.ptr, self.d_nrm.ptr, self.m,
                                               self.d_idx.ptr, self.d_d2.ptr, float(max_d2), float(trim_q), float(gate_scale),
                                               self.d_sums.ptr))
        return self.d_sums.download(np.float64, PLANE_SUMS)

    def iterate(self, n_iters, trim_q=0.5, gate_scale=20.0, max_d2=-1.0):
        c = self.ctx
        L.check(c.lib.r3d_icp_iterate_plane(c.handle, self.index.handle, self.d_src0.ptr, self.d_src.ptr, self.n, self.d_nrm.ptr,
                                            self.d_idx.ptr, self.d_d2.ptr, int(n_iters), float(trim_q), float(gate_scale),
                                            float(max_d2), self.d_state.ptr))

    def state(self):
        st = self.d_state.download(np.float64, STATE_DOUBLES)
        it = int(st[32])
        return {"T_total": st[0:16].reshape(4, 4).copy(), "T_step": st[16:32].reshape(4, 4).copy(), "iterations": it,
                "degenerate": bool(st[33]), "rms": float(st[34]), "pairs": float(st[35]),
                "rms_history": st[STATE_HISTORY:STATE_HISTORY + min(it, STATE_DOUBLES - STATE_HISTORY)].tolist()}

    def normals(self):
        return self.d_nrm.download(np.float32, self.m * 3).reshape(-1, 3)

    def free(self):
        self.index.close()
        self.ctx.sync()
        self._arena.free()


def icp_point_to_plane(src, tgt, tgt_shape=None, tgt_normals=None, init=None, max_iter=60, trim=0.5, gate_scale=20.0,
                       max_dist=None, tol=1e-7, check_every=6, max_jump=0.05, ctx=None):
    """RIGID transform (4x4, no scale) that maps the single-view cloud `src` onto the partially overlapping single-view cloud
    `tgt` -- the T_data.txt of transfer_T_icp.py:99-108 for ./point/24.txt (src) against ./point/0.txt (tgt).
    Returns (T, info).

    tgt is ORGANISED: tgt_shape = (H, W), rows in gentxtcord's raster order (p2c:34-44); its normals come from the raster
    neighbours on the GPU (or pass tgt_normals for an unorganised target).  Source rows at the camera origin (Z = 0 pixels, which
    the reference emits like any other) and non-finite rows are left out.
    init: 4x4 rough pose (e.g. the relative COLMAP pose with a guessed scale); None = identity.
    Every iteration (all on the GPU, no host round trip inside a block of `check_every`): exact nearest neighbours through the
    culled index -> residual r = n . (p - q) of every pair whose target has a plane -> per direction class of the target
    normals the `trim` order statistic of r^2, pairs above gate_scale x that are left out (non-overlap / outlier rejection that
    cannot silence a whole wall) -> 29 sums -> 6x6 solve -> move.  Stops when a step turns by less than `tol` rad and moves by
    less than `tol` x the cloud's size, or after max_iter.
    max_dist: pairs farther apart than this (point to point) never take part."""
    src = np.ascontiguousarray(src, dtype=np.float32).reshape(-1, 3)
    # rows at the camera origin (Z = 0 pixels) and non-finite rows are no points: they are dropped on the GPU (PlaneIcpDevice
    # drop_invalid: marked, sorted behind the valid rows, counted) -- the host pass that did it was 4 of the 10.7 ms of a 480x640
    # registration
    T0 = np.eye(4) if init is None else np.array(init, dtype=np.float64).reshape(4, 4)
    dev = PlaneIcpDevice(src, tgt, tgt_shape, tgt_normals, max_jump, ctx, init=None if init is None else T0, drop_invalid=True)
    try:
        dev.state_reset()
        sample = src[::max(1, src.shape[0] // 8192)].astype(np.float64)          # the cloud's size, for the stopping rule only
        sample = sample[np.isfinite(sample).all(axis=1) & (sample != 0).any(axis=1)]
        extent = (float(np.sqrt(((sample - sample.mean(0)) ** 2).sum(axis=1).mean())) if sample.shape[0] else 0.0) or 1.0
        max_d2 = -1.0 if max_dist is None else float(max_dist) ** 2
        done, converged = 0, None
        while done < max_iter and converged is None:
            k = min(max(int(check_every), 1), max_iter - done)
            dev.iterate(k, trim, gate_scale, max_d2)
            done += k
            st = dev.state()
            Ts = st["T_step"]
            ang = float(np.arccos(np.clip((np.trace(Ts[:3, :3]) - 1.0) / 2.0, -1.0, 1.0)))
            if st["degenerate"]:
                raise ValueError("point-to-plane step undefined: the matched normals leave a freedom unconstrained "
                                 "(one plane / parallel walls in the overlap), or fewer than 6 pairs")
            if ang <= tol and float(np.abs(Ts[:3, 3]).max()) <= tol * extent:
                converged = done
        info = {"iterations": st["iterations"], "rms_history": st["rms_history"], "pairs": st["pairs"], "converged_at": converged,
                "source_points_used": int(dev.n)}
        return st["T_total"] @ T0, info
    finally:
        dev.free()


def scale_from_baselines(T_icp, pose_a, pose_b):
    """The scale that takes COLMAP's (arbitrary) unit to the depth maps' unit: |t_icp| / |t_colmap| for the same image pair
    (readme.md:25).  T_icp maps camera-b points onto camera-a points in depth units; pose_* = (q_xyzw, t) rows of the COLMAP pose
    file (p_cam = R p_world + t, c2w:57-59).  Returns (scale, T_colmap_ab)."""
    from .poses import _rotation_matrix_xyzw

    def mat(q, t):
        T = np.eye(4)
        T[:3, :3], T[:3, 3] = _rotation_matrix_xyzw(q), np.asarray(t, dtype=np.float64)
        return T
    T_rel = mat(*pose_a) @ np.linalg.inv(mat(*pose_b))
    base = float(np.linalg.norm(T_rel[:3, 3]))
    if not base > 1e-12 * (float(np.linalg.norm(pose_a[1])) + float(np.linalg.norm(pose_b[1])) + 1e-300):
        raise ValueError("the two COLMAP poses share their camera centre: no baseline to compare")
    return float(np.linalg.norm(np.asarray(T_icp)[:3, 3])) / base, T_rel
