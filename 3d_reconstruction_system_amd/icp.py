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
    """4x4 T = [sR t; 0 1] minimising sum |q - (s R p + t)|^2 from the 18 sums of
    r3d_icp_accumulate: n, sum p (3), sum q (3), sum p_a q_b (9, a major), sum |p|^2, sum |q|^2."""
    sums = np.asarray(sums, dtype=np.float64)
    n = sums[0]
    if not n >= 3:
        raise ValueError("need at least 3 matched pairs (got %g)" % n)
    mu_p, mu_q = sums[1:4] / n, sums[4:7] / n
    cov_pq = sums[7:16].reshape(3, 3) / n - np.outer(mu_p, mu_q)   # E[(p-mu_p)(q-mu_q)^T]
    var_p = sums[16] / n - mu_p @ mu_p
    U, D, Vt = np.linalg.svd(cov_pq.T)                             # Sigma_qp = U D V^T
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        S[2, 2] = -1.0
    R = U @ S @ Vt
    s = float(np.trace(np.diag(D) @ S) / var_p) if with_scale else 1.0
    T = np.eye(4)
    T[:3, :3] = s * R
    T[:3, 3] = mu_q - s * (R @ mu_p)
    return T


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

    def sort_cloud(self, d_xyz_ptr, n, d_perm_ptr=None):
        L.check(self.ctx.lib.r3d_nn_index_sort_cloud(self.handle, d_xyz_ptr, int(n), d_perm_ptr))

    def close(self):
        if self.handle:
            self.ctx.lib.r3d_nn_index_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


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
        c = self.ctx
        self.d_src = c.alloc(max(src.nbytes, 16)).upload(src)
        self.d_tgt = c.alloc(tgt.nbytes).upload(tgt)
        self.d_idx = c.alloc(max(self.n * 4, 16))
        self.d_d2 = c.alloc(max(self.n * 4, 16))
        self.index = NNIndex(c, self.d_tgt.ptr, self.m) if culled else None
        self.d_perm = None
        if self.index is not None and self.n:
            # put the source cloud into the index's Morton order ONCE: rigid / similarity moves keep every
            # workgroup's 256 sources a compact blob, so no later query needs to sort.  d_perm maps back.
            self.d_perm = c.alloc(self.n * 4)
            self.index.sort_cloud(self.d_src.ptr, self.n, self.d_perm.ptr)

    def nn(self, want_stats=False):
        c = self.ctx
        if self.index is not None:
            return self.index.query(self.d_src.ptr, self.n, self.d_idx.ptr, self.d_d2.ptr, want_stats, presorted=True)
        L.check(c.lib.r3d_icp_nn(c.handle, self.d_src.ptr, self.n, self.d_tgt.ptr, self.m, self.d_idx.ptr,
                                 self.d_d2.ptr))
        return 0

    def sums(self, max_d2=-1.0):
        c = self.ctx
        out = np.zeros(18, dtype=np.float64)
        L.check(c.lib.r3d_icp_accumulate(c.handle, self.d_src.ptr, self.n, self.d_tgt.ptr, self.m, self.d_idx.ptr,
                                         self.d_d2.ptr if max_d2 >= 0 else None, float(max_d2), out.ctypes.data))
        return out

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
        if self.index is not None:
            self.index.close()
        for b in (self.d_src, self.d_tgt, self.d_idx, self.d_d2, self.d_perm):
            if b is not None:
                b.free()


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


def icp_similarity(src, tgt, max_iter=30, tol=1e-7, with_scale=True, trim_d2=None, ctx=None, culled=True):
    """Iterate NN + Umeyama until the RMS match distance stops improving by more than `tol`
    (relative).  Returns (T 4x4 mapping src -> tgt, info dict).  trim_d2: ignore pairs whose
    squared distance exceeds it (None = use all)."""
    dev = IcpDevice(src, tgt, ctx, culled)
    T_total = np.eye(4)
    history = []
    prev = None
    try:
        for it in range(max_iter):
            dev.nn()
            sums = dev.sums(-1.0 if trim_d2 is None else float(trim_d2))
            n = sums[0]
            # sum |p-q|^2 = sum|p|^2 + sum|q|^2 - 2 tr(sum p q^T)
            rms = float(np.sqrt(max(sums[16] + sums[17] - 2.0 * (sums[7] + sums[11] + sums[15]), 0.0) / max(n, 1.0)))
            history.append(rms)
            T = umeyama_from_sums(sums, with_scale)
            dev.move_source(T)
            T_total = T @ T_total
            if prev is not None and abs(prev - rms) <= tol * max(prev, 1e-30):
                break
            prev = rms
        dev.ctx.sync()
    finally:
        dev.free()
    return T_total, {"iterations": len(history), "rms_history": history}
