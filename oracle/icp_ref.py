"""Oracle (test infrastructure): ICP estimation restated on the CPU.  PARITY UNPINNED.

The reference repo has no ICP code to pin against (other_tools/transfer_T_icp.py:99-108 only
reads a T_data.txt made by an external tool; requirements.txt:52-53 pins open3d but nothing
imports it).  This file states the build's own definition:
  * nearest neighbour under squared L2 evaluated in fp32 as fma(dz,dz, fma(dy,dy, dx*dx)),
    lowest target index wins exact ties;
  * the 18 pair sums in fp64;
  * Umeyama (IEEE PAMI 13(4), 1991) closed-form similarity from point pairs.
Anchors: synthetic clouds with a known (s, R, t) and exact correspondences; scipy's cKDTree as
an independent NN cross-check (fp64 distances).
"""
import numpy as np


def _fma32(a, b, c):
    """round32(a*b + c) for float32 arrays: the product of two fp32 numbers is exact in fp64; the
    fp64 sum is then rounded to fp32 (double rounding can differ from a true fma only when the
    fp64 sum lands exactly on an fp32 tie, ~2^-29 of cases)."""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def pair_d2(src, tgt):
    """[N,M] float32 squared distances with the kernel's exact expression."""
    s = np.asarray(src, dtype=np.float32)[:, None, :]
    t = np.asarray(tgt, dtype=np.float32)[None, :, :]
    dx, dy, dz = s[..., 0] - t[..., 0], s[..., 1] - t[..., 1], s[..., 2] - t[..., 2]
    return _fma32(dz, dz, _fma32(dy, dy, dx * dx))


def nearest_neighbours(src, tgt, chunk=512):
    """(idx uint32 [N], d2 float32 [N]); np.argmin returns the lowest index among ties."""
    src = np.asarray(src, dtype=np.float32)
    tgt = np.asarray(tgt, dtype=np.float32)
    idx = np.empty(src.shape[0], dtype=np.uint32)
    d2 = np.empty(src.shape[0], dtype=np.float32)
    for lo in range(0, src.shape[0], chunk):
        d = pair_d2(src[lo:lo + chunk], tgt)
        k = np.argmin(d, axis=1)
        idx[lo:lo + chunk] = k
        d2[lo:lo + chunk] = d[np.arange(d.shape[0]), k]
    return idx, d2


def pair_weights(d2, dead_zone):
    """w = max(0, 1 - dead_zone/d), d = sqrt(d2) in fp64: the IRLS weight of the cost max(0, d - dead_zone)^2.
    dead_zone <= 0: all ones."""
    d2 = np.asarray(d2, dtype=np.float32).astype(np.float64)
    if not dead_zone > 0:
        return np.ones_like(d2)
    d = np.sqrt(d2)
    dz = float(np.float32(dead_zone))
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.where(d > dz, 1.0 - dz / d, 0.0)


def pair_sums(src, tgt, idx, d2=None, max_d2=-1.0, dead_zone=0.0):
    """The 18 fp64 sums: n, sum p, sum q, sum p_a q_b (a major), sum |p|^2, sum |q|^2; weighted by
    pair_weights(d2, dead_zone) when dead_zone > 0 (n is then the weight total)."""
    p = np.asarray(src, dtype=np.float32).astype(np.float64)
    q = np.asarray(tgt, dtype=np.float32).astype(np.float64)[np.asarray(idx, dtype=np.int64)]
    w = pair_weights(d2, dead_zone) if dead_zone > 0 else np.ones(p.shape[0])
    if max_d2 >= 0:
        keep = np.asarray(d2) <= np.float32(max_d2)
        p, q, w = p[keep], q[keep], w[keep]
    finite = np.isfinite(p).all(1) & np.isfinite(q).all(1)         # a row with a NaN / inf coordinate is no pair
    p, q, w = p[finite], q[finite], w[finite]
    out = np.zeros(18)
    wp = p * w[:, None]
    out[0] = w.sum()
    out[1:4] = wp.sum(0)
    out[4:7] = (q * w[:, None]).sum(0)
    out[7:16] = (wp[:, :, None] * q[:, None, :]).sum(0).reshape(9)
    out[16] = (wp * p).sum()
    out[17] = (q * q * w[:, None]).sum()
    return out


def swap_pair_sums(s):
    out = np.array(s, dtype=np.float64)
    out[1:4], out[4:7] = s[4:7], s[1:4]
    out[7:16] = np.asarray(s[7:16]).reshape(3, 3).T.reshape(9)
    out[16], out[17] = s[17], s[16]
    return out


def umeyama_from_sums(sums, with_scale=True):
    """Umeyama's closed form from the 18 (weighted) sums, numpy SVD."""
    sums = np.asarray(sums, dtype=np.float64)
    n = sums[0]
    mu_p, mu_q = sums[1:4] / n, sums[4:7] / n
    cov_pq = sums[7:16].reshape(3, 3) / n - np.outer(mu_p, mu_q)
    var_p = sums[16] / n - mu_p @ mu_p
    U, D, Vt = np.linalg.svd(cov_pq.T)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        S[2, 2] = -1.0
    R = U @ S @ Vt
    s = float(np.trace(np.diag(D) @ S) / var_p) if with_scale else 1.0
    T = np.eye(4)
    T[:3, :3] = s * R
    T[:3, 3] = mu_q - s * (R @ mu_p)
    return T


def umeyama(p, q, with_scale=True):
    """T = [sR t; 0 1] minimising sum |q_i - (s R p_i + t)|^2, straight from the point pairs."""
    p = np.asarray(p, dtype=np.float64)
    q = np.asarray(q, dtype=np.float64)
    mp, mq = p.mean(0), q.mean(0)
    pc, qc = p - mp, q - mq
    sigma = qc.T @ pc / p.shape[0]
    U, D, Vt = np.linalg.svd(sigma)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        S[2, 2] = -1
    R = U @ S @ Vt
    s = np.trace(np.diag(D) @ S) / (pc ** 2).sum(1).mean() if with_scale else 1.0
    T = np.eye(4)
    T[:3, :3] = s * R
    T[:3, 3] = mq - s * R @ mp
    return T


def icp_similarity(src, tgt, max_iter=30, tol=1e-7, with_scale=True):
    src = np.asarray(src, dtype=np.float32).copy()
    tgt = np.asarray(tgt, dtype=np.float32)
    T_total = np.eye(4)
    prev = None
    for _ in range(max_iter):
        idx, _d2 = nearest_neighbours(src, tgt)
        q = tgt[idx.astype(np.int64)].astype(np.float64)
        p = src.astype(np.float64)
        rms = float(np.sqrt(((p - q) ** 2).sum(1).mean()))
        T = umeyama(p, q, with_scale)
        src = (p @ T[:3, :3].T + T[:3, 3]).astype(np.float32)
        T_total = T @ T_total
        if prev is not None and abs(prev - rms) <= tol * max(prev, 1e-30):
            break
        prev = rms
    return T_total


def apply_T32(xyz, T):
    """float32 cloud moved by a 4x4 in fp64, rounded once (what r3d_apply_T does)."""
    x = np.asarray(xyz, dtype=np.float32).astype(np.float64)
    return (x @ T[:3, :3].T + T[:3, 3]).astype(np.float32)


def quantile_lower(values, q):
    """Element of rank floor(q (m - 1)) of the m finite values (r3d_select_quantile_f32's rule); -1 when there are none."""
    v = np.asarray(values, dtype=np.float32)
    v = np.sort(v[np.isfinite(v)])
    return float(v[int(np.floor(q * (v.size - 1)))]) if v.size else -1.0


def target_spacing(tgt, max_queries=32768):
    """Median NN distance between the interleaved halves of the target (its sampling resolution)."""
    tgt = np.asarray(tgt, dtype=np.float32)
    base, probe = tgt[1::2], tgt[0::2]
    probe = probe[::max(1, probe.shape[0] // max_queries)]
    _i, d2 = nearest_neighbours(probe, base)
    d2 = np.sort(d2[np.isfinite(d2)])          # the "lower" median: rank floor(0.5 (m - 1)), like the GPU's selection
    return float(np.sqrt(d2[(d2.size - 1) // 2])) if d2.size else 0.0


INIT_SAMPLES, INIT_KEEP, INIT_PREFER_MOMENTS = 8192, 0.8, 1.05


def init_candidates(P, Q, with_scale=True):
    """icp.init_candidates restated from the points: the moments start and the four proper principal-axis alignments."""
    mu_p, mu_q = P.mean(0), Q.mean(0)
    cov_p, cov_q = (P - mu_p).T @ (P - mu_p) / P.shape[0], (Q - mu_q).T @ (Q - mu_q) / Q.shape[0]
    r_p, r_q = np.sqrt(np.trace(cov_p)), np.sqrt(np.trace(cov_q))
    s = (r_q / r_p if r_p > 0 and r_q > 0 else 1.0) if with_scale else 1.0

    def make(Rm):
        T = np.eye(4)
        T[:3, :3] = s * Rm
        T[:3, 3] = mu_q - s * (Rm @ mu_p)
        return T

    out = [make(np.eye(3))]
    wp, Vp = np.linalg.eigh(0.5 * (cov_p + cov_p.T))
    wq, Vq = np.linalg.eigh(0.5 * (cov_q + cov_q.T))
    if wp[2] <= 0 or wq[2] <= 0:
        return out
    for perm in ((0, 1, 2), (0, 2, 1), (1, 0, 2), (1, 2, 0), (2, 0, 1), (2, 1, 0)):
        Pm = np.zeros((3, 3))
        for a in range(3):
            Pm[a, perm[a]] = 1.0
        for sx in (1.0, -1.0):
            for sy in (1.0, -1.0):
                Rm = Vq @ np.diag([sx, sy, 1.0]) @ Pm @ Vp.T
                if np.linalg.det(Rm) < 0:
                    Rm = Vq @ np.diag([sx, sy, -1.0]) @ Pm @ Vp.T
                out.append(make(Rm))
    return out


def trimmed_mean(d2, keep=0.8):
    """Mean (fp64) of the finite values that are <= their `keep` order statistic ("lower" rule: rank floor(keep (m - 1)));
    +inf when there are none (r3d_trimmed_means_f32)."""
    v = np.asarray(d2, dtype=np.float32)
    v = v[np.isfinite(v)]
    if v.size == 0:
        return float("inf")
    gate = np.sort(v)[int(np.floor(keep * (v.size - 1)))]
    return float(v[v <= gate].astype(np.float64).mean())


def init_costs(src, tgt, transforms):
    """icp.IcpDevice.init_costs restated: symmetric trimmed misfit of each start on strided samples."""
    ss = src[::max(1, src.shape[0] // INIT_SAMPLES)]
    st = tgt[::max(1, tgt.shape[0] // INIT_SAMPLES)]
    costs = []
    for T in transforms:
        s2 = float(np.cbrt(abs(np.linalg.det(T[:3, :3])))) ** 2
        fwd = trimmed_mean(nearest_neighbours(apply_T32(ss, T), tgt)[1], INIT_KEEP)
        back = trimmed_mean(nearest_neighbours(apply_T32(st, np.linalg.inv(T)), src)[1], INIT_KEEP) * s2
        costs.append(fwd + back)
    return costs


def icp_similarity_auto(src, tgt, max_iter=60, tol=1e-7, with_scale=True, dead_zone=None, max_coarse=40,
                        coarse_tol=2e-4, trim=None, check_every=4):
    """The build's full estimator restated: moments init -> symmetric dead-zone ICP -> plain ICP
    (3d_reconstruction_system_amd/icp.py: icp_similarity(init="auto")).  Brute-force NN: small clouds only."""
    src = np.asarray(src, dtype=np.float32)
    tgt = np.asarray(tgt, dtype=np.float32)
    P, Q = src.astype(np.float64), tgt.astype(np.float64)
    mu_p, mu_q = P.mean(0), Q.mean(0)
    r_p, r_q = np.sqrt(((P - mu_p) ** 2).sum(1).mean()), np.sqrt(((Q - mu_q) ** 2).sum(1).mean())
    T_total = np.eye(4)
    s0 = r_q / r_p if with_scale else 1.0
    T_total[:3, :3] *= s0
    T_total[:3, 3] = mu_q - s0 * mu_p
    choice = 0
    if min(src.shape[0], tgt.shape[0]) >= 16:
        cands = init_candidates(P, Q, with_scale)
        costs = init_costs(src, tgt, cands)
        choice = int(np.argmin(costs))
        if costs[0] <= INIT_PREFER_MOMENTS * costs[choice]:
            choice = 0
        T_total = cands[choice]
    cur = apply_T32(src, T_total)
    d0 = float(dead_zone) if dead_zone is not None else 2.0 * target_spacing(tgt)
    snap = np.asarray(src, dtype=np.float32)      # the back search runs against the source AS GIVEN (icp.IcpDevice.source_index);
    T_since = T_total.copy()                        # T_since = everything the source has been moved by since
    coarse = 0
    for _ in range(max_coarse):
        ia, da = nearest_neighbours(cur, tgt)
        ct = None if trim is None else 1.0 - (1.0 - trim) / 4.0      # the coarse stage trims more gently (icp.py: coarse_trim)
        ga = quantile_lower(da, ct) if ct is not None and ct < 1.0 else -1.0
        sums = pair_sums(cur, tgt, ia, da, ga, d0)
        T_inv = np.linalg.inv(T_since)
        s_since = float(np.cbrt(abs(np.linalg.det(T_since[:3, :3]))))
        moved = apply_T32(tgt, T_inv)
        ib, db = nearest_neighbours(moved, snap)
        gb = quantile_lower(db, ct) if ct is not None and ct < 1.0 else -1.0
        sums = sums + swap_pair_sums(pair_sums(tgt, cur, ib, db, gb, d0 / s_since))
        if not sums[0] >= 3.0:
            break
        T = umeyama_from_sums(sums, with_scale)
        cur = apply_T32(cur, T)
        T_total = T @ T_total
        T_since = T @ T_since
        coarse += 1
        step = max(np.abs(T[:3, :3] - np.eye(3)).max(), np.abs(T[:3, 3]).max() / r_q)
        if step <= coarse_tol:
            break
    prev = None
    fine = 0
    gate = -1.0
    for it in range(max_iter):
        idx, d2 = nearest_neighbours(cur, tgt)
        if trim is not None and trim < 1.0 and it % check_every == 0:     # re-ranked once per block, like the GPU loop
            gate = quantile_lower(d2, trim)
        sums = pair_sums(cur, tgt, idx, d2, gate)
        rms = float(np.sqrt(max(sums[16] + sums[17] - 2 * (sums[7] + sums[11] + sums[15]), 0.0) / sums[0]))
        T = umeyama_from_sums(sums, with_scale)
        cur = apply_T32(cur, T)
        T_total = T @ T_total
        fine += 1
        if prev is not None and abs(prev - rms) <= tol * max(prev, 1e-30):
            break
        prev = rms
    return T_total, {"coarse_iterations": coarse, "iterations": fine, "dead_zone": d0, "init_choice": choice}


def synthetic_pair(n_tgt=4000, n_src=3000, seed=7, s=1.7, angle_deg=10.0, t_norm=0.5, noise=0.0, extent=20.0):
    """SURVEY 8(d) C3 recipe at a chosen size: tgt uniform in a cube (+ optional noise);
    src = the inverse similarity applied to a subset of tgt, so the transform that maps src back
    onto tgt is exactly (s, R, t)."""
    rng = np.random.default_rng(seed)
    tgt = rng.random((n_tgt, 3)) * extent
    axis = rng.normal(size=3)
    axis /= np.linalg.norm(axis)
    a = np.deg2rad(angle_deg)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    R = np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * K @ K
    t = rng.normal(size=3)
    t *= t_norm / np.linalg.norm(t)
    T = np.eye(4)
    T[:3, :3] = s * R
    T[:3, 3] = t
    pick = rng.permutation(n_tgt)[:n_src]
    q = tgt[pick] + rng.normal(size=(n_src, 3)) * noise
    src = (q - t) @ np.linalg.inv(s * R).T
    return src.astype(np.float32), tgt.astype(np.float32), T, pick
