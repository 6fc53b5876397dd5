"""bench.py --workload apply | icp | voxel | c5: one JSON line for a secondary kernel (single GPU), and config 5 over the ranks of a launcher."""
import importlib
import json
import os
import sys
import tempfile
import time

import numpy as np

from bench_common import BYTES_PER_POINT, FRAMES_PER_GPU, H, HBM_COPY_GBS, HBM_PEAK_GBS, ROOT, W, cpu_model, run_child  # noqa: F401


def apply_cpu_baseline(sample_points=200000):
    """Loop-faithful restatement of transfer_T_icp.py:71-97 (local_world with flag=True: per-line parse, np.dot(T, p), three
    list appends, one text line out), 1 core, on a bounded sample.  Reported, not optimised against."""
    from oracle import fusion_ref as O
    rng = np.random.default_rng(1234)
    pts = rng.normal(size=(sample_points, 3)) * 50
    T = np.eye(4)
    T[:3, :3] *= 1.7
    T[:3, 3] = (1, 2, 3)
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "24.txt")
        with open(src, "w") as f:
            for x, y, z in pts.tolist():
                f.write("%r,%r,%r\n" % (x, y, z))
        xs, ys, zs = [], [], []
        t0 = time.perf_counter()
        with open(os.path.join(td, "world.txt"), "w") as fout:
            O.local_world_loop(src, fout, T, xs, ys, zs, True)
        dt = time.perf_counter() - t0
    return {"value": round(sample_points / dt / 1e6, 5), "unit": "Mpoints/s", "cores": 1, "kind": "port", "cpu_model": cpu_model(),
            "sample": "%d points through local_world's per-line loop (parse, 4x4 dot, text out) as transfer_T_icp.py:71-97; "
                      "%.1f s" % (sample_points, dt), "host_cpus": os.cpu_count()}


def secondary(a):
    """One JSON line for a secondary kernel (single GPU, HIP-event stopwatch of the library on its own stream)."""
    r3d = importlib.import_module("3d_reconstruction_system_amd")
    ctx = r3d.Context(0)
    rng = np.random.default_rng(1234)

    def timed(fn, iters):
        """Median of 5 groups of `iters`/5 launches, after a warm-up of >= 5 launches and >= 60 ms (the clocks of an
        idle GPU boost for the first ~2 ms and then dip for ~20 ms: profiles/r02_*: neither belongs in a rate)."""
        t0 = time.perf_counter()
        k = 0
        while k < 5 or time.perf_counter() - t0 < 0.06:
            fn()
            k += 1
            if k % 5 == 0:
                ctx.sync()
        ctx.sync()
        per = max(iters // 5, 1)
        groups = []
        for _ in range(5):
            ctx.timer_start()
            for _ in range(per):
                fn()
            groups.append(ctx.timer_stop() / per)
        return sorted(groups)[2]

    if a.workload == "apply":
        n = FRAMES_PER_GPU * H * W
        d_in = ctx.alloc(n * 12).upload((rng.normal(size=(n, 3)) * 50).astype(np.float32))
        d_out = ctx.alloc(n * 12)
        T = np.eye(4)
        T[:3, :3] *= 1.7
        T[:3, 3] = (1, 2, 3)
        ms = timed(lambda: r3d.apply_T_device(ctx, d_in.ptr, np.float32, n, T, d_out.ptr, np.float32), max(a.steps // 10, 20))
        gbs = n * 24 / ms / 1e6
        line = {"metric": "Mpoints/s apply-T (4x4 on a 49.2 Mpoint f32 cloud)", "value": round(n / ms / 1e3, 1), "unit": "Mpoints/s",
                "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None, "kernel": "apply_lane_kernel<f32,affine>",
                             "frac_of_measured_copy": round(gbs / HBM_COPY_GBS, 4),
                             "kernel_ms": round(ms, 5), "algorithmic_bytes_per_launch": n * 24}}
        if not a.no_cpu_baseline:
            line["cpu_baseline"] = apply_cpu_baseline()
    elif a.workload == "icp":
        icp = importlib.import_module("3d_reconstruction_system_amd.icp")
        m = 500000
        # SURVEY.md 8(d) C3 recipe: target uniform in a 20 m cube + N(0, 0.01); source = inverse similarity
        # (s=1.7, 10 degrees, |t|=0.5) of a permutation of the noise-free target; no initial guess
        tgt0 = rng.random((m, 3)) * 20
        ax = rng.normal(size=3)
        ax /= np.linalg.norm(ax)
        ang = np.deg2rad(10.0)
        K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
        Rm = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K
        tv = rng.normal(size=3)
        tv *= 0.5 / np.linalg.norm(tv)
        T_true = np.eye(4)
        T_true[:3, :3], T_true[:3, 3] = 1.7 * Rm, tv
        src = ((tgt0[rng.permutation(m)] - tv) @ np.linalg.inv(1.7 * Rm).T).astype(np.float32)
        tgt = (tgt0 + rng.normal(size=tgt0.shape) * 0.01).astype(np.float32)
        icp.icp_similarity(src[:3000], tgt[:3000], max_iter=2, ctx=ctx)                      # warm-up
        walls = []
        for _ in range(4):   # the first full-size call also grows the library's scratch buffers (hipMalloc): reported apart
            t0 = time.perf_counter()
            T, info = icp.icp_similarity(src, tgt, ctx=ctx)
            walls.append((time.perf_counter() - t0) * 1e3)
        first_call_ms, wall_ms = walls[0], sorted(walls[1:])[1]
        near = (src.astype(np.float64) @ (T_true[:3, :3] * 1.002).T + T_true[:3, 3]).astype(np.float32)
        dev_b = icp.IcpDevice(near, tgt, ctx, culled=False)
        ms_b = timed(dev_b.nn, 3)
        dev_b.free()
        dev_c = icp.IcpDevice(near, tgt, ctx, culled=True)
        ms_c = timed(dev_c.nn, 20)
        dev_c.state_reset()
        ms_it = timed(lambda: dev_c.iterate(6), 10) / 6     # as the estimator runs them: six per enqueue (the later five start warm)
        dev_c.free()
        tf = m * m * 8 / ms_b / 1e9
        # the reference's own case (readme.md:25): two partially overlapping 480x640 single views, rigid point-to-plane ICP
        Sy = importlib.import_module("3d_reconstruction_system_amd.synthetic")
        v2 = Sy.two_views(480, 640, yaw_deg=15.0, baseline=(0.35, 0.05, -0.2), depth_noise=0.001, seed=1)
        pa = r3d.unproject(v2["depth_a"], v2["K"], ctx=ctx)
        pb = r3d.unproject(v2["depth_b"], v2["K"], ctx=ctx)
        ca, sa = np.cos(np.deg2rad(5.0)), np.sin(np.deg2rad(5.0))
        E = np.eye(4)
        E[:3, :3] = [[ca, 0, sa], [0, 1, 0], [-sa, 0, ca]]
        E[:3, 3] = (0.06, -0.05, 0.06)
        T0 = E @ v2["T_ab"]
        icp.icp_point_to_plane(pb[:60000], pa, tgt_shape=(480, 640), init=T0, max_iter=2, ctx=ctx)      # warm-up
        plane_walls = []
        for _ in range(4):   # (first full-size call apart, as above)
            t0 = time.perf_counter()
            Tp, infop = icp.icp_point_to_plane(pb, pa, tgt_shape=(480, 640), init=T0, ctx=ctx)
            plane_walls.append((time.perf_counter() - t0) * 1e3)
        plane_ms = sorted(plane_walls[1:])[1]
        devp = icp.PlaneIcpDevice(pb, pa, (480, 640), ctx=ctx)
        devp.move_source(Tp)
        devp.state_reset()
        ms_pit = timed(lambda: devp.iterate(6), 10) / 6
        devp.free()
        plane = {"what": "two 480x640 single views of a room, 15 deg apart, 67 % overlap, depth noise 0.1 %, start 5 deg / 10 cm off",
                 "wall_ms": round(plane_ms, 2), "first_call_ms": round(plane_walls[0], 2), "iterations": infop["iterations"], "iteration_ms": round(ms_pit, 4),
                 "T_error_max_abs": float(np.abs(Tp - v2["T_ab"]).max()), "pairs": infop["pairs"]}
        line = {"metric": "ICP similarity estimation, two 500k-point clouds (C3: s=1.7, 10 deg, |t|=0.5, no initial guess)",
                "value": round(wall_ms, 2), "unit": "ms wall (upload, index builds, coarse + fine stages; median of 3 calls after the first)",
                "first_call_ms": round(first_call_ms, 2), "higher_is_better": False,
                "T_error_max_abs": float(np.abs(T - T_true).max()), "coarse_iterations": info["coarse_iterations"],
                "fine_iterations": info["iterations"], "final_rms": info["rms_history"][-1],
                "fine_iteration_ms": round(ms_it, 4), "culled_nn_ms": round(ms_c, 4), "bruteforce_nn_ms": round(ms_b, 3),
                "point_to_plane_two_views": plane,
                "roofline": {"bound": "valu", "achieved": round(tf, 1), "peak": 157.3, "unit": "TFLOP/s", "frac": round(tf / 157.3, 4),
                             "traffic": None, "kernel": "nn_kernel<4> (brute force, 8 flop/pair)", "kernel_ms": round(ms_b, 3)}}
    elif a.workload == "c5":
        # BASELINE config 5 geometry per GPU: AirSim 1920x1080 f32 depth + RGB, fused cloud carrying colour + voxel insert
        V = importlib.import_module("3d_reconstruction_system_amd.voxelmap")
        F, h5, w5 = max(1, min(a.frames, 100)) if a.frames != FRAMES_PER_GPU else 50, 1080, 1920
        n = F * h5 * w5
        # (poses first, then depth, then colour: a checker can regenerate the first k frames without drawing all F)
        tab = r3d.pose_table(rng.normal(size=(F, 4)), rng.normal(size=(F, 3)) * 10)
        depth = (rng.random((F, h5, w5), dtype=np.float32) * 99.5 + 0.5)
        rgb = rng.integers(0, 256, size=(F, h5, w5, 3), dtype=np.uint8)
        d_depth, d_rgb, d_pose = ctx.alloc(depth.nbytes).upload(depth), ctx.alloc(rgb.nbytes).upload(rgb), ctx.alloc(tab.nbytes).upload(tab)
        del depth, rgb
        d_xyz, d_rgba = ctx.alloc(n * 12), ctx.alloc(n * 4)
        cam = ctx.camera(h5, w5, 960.0, 960.0, 959.5, 539.5)
        ms = timed(lambda: r3d.fuse_frames_rgb_device(ctx, cam, d_depth.ptr, np.float32, F, d_pose.ptr, d_rgb.ptr, d_xyz.ptr,
                                                      np.float32, d_rgba.ptr), max(a.steps // 20, 20))
        vs = V.VoxelSet(0.1, 2 * n, ctx)

        def both():
            vs.clear()
            vs.insert_device(d_xyz.ptr, n)
        ms_clear = timed(vs.clear, 5)
        ms_v = timed(both, 5) - ms_clear
        both()
        st_all = vs.stats()

        def one_launch():   # the cloud and the map from one kernel (r3d_fuse_frames_voxel): the cloud is not read back
            vs.clear()
            r3d.fuse_frames_voxel_device(ctx, cam, d_depth.ptr, np.float32, F, d_pose.ptr, d_rgb.ptr, d_xyz.ptr, d_rgba.ptr, vs)
        ms_one = timed(one_launch, 5) - ms_clear
        one_launch()
        st_one = vs.stats()
        # a checkable digest of the map's voxel half: the occupied set of the first k frames' cloud (the test suite forms the
        # same set with the oracle and compares count, ignored points and two order-independent digests of the codes)
        k_chk = min(F, 3)
        vs.clear()
        vs.insert_device(d_xyz.ptr, k_chk * h5 * w5)
        st_k = vs.stats()
        codes = vs.codes()
        voxel_check = {"frames": k_chk, "points": k_chk * h5 * w5, "voxels": int(codes.shape[0]), "ignored_points": st_k["ignored_points"],
                       "overflow": st_k["overflow"], "codes_xor": int(np.bitwise_xor.reduce(codes)) if codes.size else 0,
                       "codes_sum_mod_2_64": int(np.sum(codes, dtype=np.uint64)) if codes.size else 0,
                       "seed": 1234, "resolution": 0.1}
        bpp = 16 + 7
        gbs = n * bpp / ms / 1e6
        line = {"metric": "Mpoints/s fused RGBD (1920x1080 f32 depth + RGB -> f32 xyz + rgba), %d frames" % F,
                "value": round(n / ms / 1e3, 1), "unit": "Mpoints/s", "voxel_insert_ms": round(ms_v, 3),
                "fuse_plus_voxel_Mpoints_s": round(n / (ms + ms_v) / 1e3, 1), "voxels": st_all["voxels"],
                "one_launch_cloud_and_voxels": {"ms": round(ms_one, 3), "Mpoints_s": round(n / ms_one / 1e3, 1),
                                                "same_counters_as_two_calls": st_one == st_all},
                "voxel_check": voxel_check,
                "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None, "kernel": "fuse_rgb_kernel<f32,pose>",
                             "frac_of_measured_copy": round(gbs / HBM_COPY_GBS, 4),
                             "kernel_ms": round(ms, 5), "algorithmic_bytes_per_launch": n * bpp,
                             "bytes_per_point": "16 (f32 depth in, f32 xyz out) + 7 (rgb in, rgba out)"}}
    else:
        V = importlib.import_module("3d_reconstruction_system_amd.voxelmap")
        F = FRAMES_PER_GPU
        n = F * H * W
        depth = rng.integers(1, 256, size=(F, H, W), dtype=np.uint8)
        tab = r3d.pose_table(rng.normal(size=(F, 4)), rng.normal(size=(F, 3)))
        d_depth, d_pose, d_xyz = ctx.alloc(n).upload(depth), ctx.alloc(tab.nbytes).upload(tab), ctx.alloc(n * 12)
        cam = ctx.camera(H, W, *r3d.REF_INTRINSICS)
        r3d.fuse_frames_device(ctx, cam, d_depth.ptr, np.uint8, F, d_pose.ptr, d_xyz.ptr, np.float32)
        vs = V.VoxelSet(0.1, 2 * n, ctx)

        def both():
            vs.clear()
            vs.insert_device(d_xyz.ptr, n)
        ms_clear = timed(vs.clear, 10)
        per_path = {}
        for label, path in (("cas_lds_set", 1), ("sort_merge", 2), ("auto", 0)):
            ctx.set_tuning("voxel_path", path)
            per_path[label] = {"ms": round(timed(both, 10) - ms_clear, 4), "path_taken": ctx.get_tuning("voxel_last_path")}
        ms = per_path["auto"]["ms"]
        both()
        st = vs.stats()
        # algorithmic bytes of a set insert: every point read once (12 B), every distinct voxel written once (8 B)
        alg = n * 12 + st["voxels"] * 8
        gbs = alg / ms / 1e6
        # HBM bytes of one insert: a RECORDED figure (rocprofv3 --pmc passes over tools/voxel_sort_once.py, the same cloud and table:
        # tools/collect_voxel_profile.sh), given only when that profile's cloud is this one
        traffic, traffic_source = None, None
        try:
            import glob
            rec_path = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_voxel_stage_pmc.json")))[-1]
            rec = json.load(open(rec_path))
            if rec.get("algorithmic_bytes") == alg:
                traffic = int(rec["hbm_bytes_per_insert"])
                traffic_source = "recorded: %s (FETCH_SIZE x 2 + WRITE_SIZE per stage)" % os.path.relpath(rec_path, ROOT)
        except Exception:
            pass
        line = {"metric": "Mpoints/s voxel insert (C2 cloud, 0.1 m, worst case ~1 voxel per point)", "value": round(n / ms / 1e3, 1),
                "unit": "Mpoints/s", "voxels": st["voxels"], "kernel_ms": round(ms, 4), "paths": per_path,
                "table_slots": 1 << int(np.ceil(np.log2(2 * n))),
                "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                             "algorithmic_bytes_per_launch": alg,
                             "kernel": "voxel_bin_kernel + segment_histogram_kernel + segment_scatter_kernel + voxel_merge32_kernel "
                                       "(sort-merge insert: 17 + 1 + 9 + 4 B/point of streams + 8 B per table slot; "
                                       "the CAS path is bound by scattered 64-bit atomics at ~19 G/s instead)",
                             "designed_bytes_per_launch": n * 31 + (1 << int(np.ceil(np.log2(2 * n)))) * 8}}
    line.setdefault("higher_is_better", True)
    line.update({"n_gpus": 1, "data": "synthetic", "dtype": "f64" if a.workload in ("apply", "c5") else "f32",
                 "config": {"workload": a.workload}})
    print(json.dumps(line), flush=True)
    ctx.close()


def c5_sharded(a):
    """BASELINE config 5's shape over the GPUs of a node, torch-free: `torch.distributed.run --nproc-per-node N bench.py
    --workload c5 --gpus N`.  Every rank owns F frames of 1920x1080 f32 depth + RGB; a step = fuse them with colour (one
    launch), voxelise the rank's shard into its own HBM hash set, unite the sets through the C ABI (r3d_voxelset_union:
    all-gather of the DISTINCT codes only, 8 B/voxel; the 16 B/point of the coloured cloud never leave their GPU).
    Synthetic depth is random, i.e. the worst case of ~1 voxel per point."""
    r3d = importlib.import_module("3d_reconstruction_system_amd")
    V = importlib.import_module("3d_reconstruction_system_amd.voxelmap")
    CM = importlib.import_module("3d_reconstruction_system_amd.comm")
    rank, world = CM.env_rank_world()
    if world != a.gpus:
        sys.exit("WORLD_SIZE=%d does not match --gpus %d" % (world, a.gpus))
    ctx = r3d.Context(CM.env_local_device())
    comm = CM.Comm.from_env(ctx)
    F = max(1, min(a.frames, 250)) if a.frames != FRAMES_PER_GPU else 50
    h5, w5 = 1080, 1920
    n = F * h5 * w5
    rng = np.random.default_rng(5 + rank)
    depth = rng.random((F, h5, w5), dtype=np.float32) * 99.5 + 0.5
    rgb = rng.integers(0, 256, size=(F, h5, w5, 3), dtype=np.uint8)
    tab = r3d.pose_table(rng.normal(size=(F, 4)), rng.normal(size=(F, 3)) * 10)
    d_depth, d_rgb, d_pose = ctx.alloc(depth.nbytes).upload(depth), ctx.alloc(rgb.nbytes).upload(rgb), ctx.alloc(tab.nbytes).upload(tab)
    del depth, rgb
    d_xyz, d_rgba = ctx.alloc(n * 12), ctx.alloc(n * 4)
    cam = ctx.camera(h5, w5, 960.0, 960.0, 959.5, 539.5)
    vs = V.VoxelSet(0.1, 2 * n * world, ctx)
    d_t = ctx.alloc(8 * (world + 1))

    def step():
        r3d.fuse_frames_rgb_device(ctx, cam, d_depth.ptr, np.float32, F, d_pose.ptr, d_rgb.ptr, d_xyz.ptr, np.float32, d_rgba.ptr)
        vs.clear()
        vs.insert_device(d_xyz.ptr, n)
        vs.union_across(comm)

    def max_over_ranks(seconds):
        mine = np.array([seconds])
        ctx.lib.r3d_memcpy_h2d(ctx.handle, d_t.ptr + 8 * world, mine.ctypes.data, 8)
        comm.allgather(d_t.ptr + 8 * world, [8] * world, d_t.ptr)
        return float(d_t.download(np.float64, world).max())

    steps, warm = max(1, min(a.steps, 50)), max(1, min(a.warmup, 5))
    for _ in range(warm):
        step()
    comm.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    comm.barrier()
    sec = max_over_ranks(time.perf_counter() - t0)
    st = vs.stats()
    ms_fuse = []
    for _ in range(5):
        ctx.timer_start()
        r3d.fuse_frames_rgb_device(ctx, cam, d_depth.ptr, np.float32, F, d_pose.ptr, d_rgb.ptr, d_xyz.ptr, np.float32, d_rgba.ptr)
        ms_fuse.append(ctx.timer_stop())
    ms = sorted(ms_fuse)[2]
    if rank == 0:
        gbs = n * 23 / ms / 1e6
        print(json.dumps({
            "metric": "Mpoints/s fused RGBD + voxel map (1920x1080 f32 depth + RGB, %d frames per GPU, one map)" % F,
            "value": round(world * n * steps / sec / 1e6, 1), "unit": "Mpoints/s", "n_gpus": world, "steps": steps, "warmup": warm,
            "ms_per_step": round(sec / steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic (random depth: ~1 voxel per point, the worst case for the map)",
            "config": {"workload": "C5: fuse with colour + voxel insert + union of the ranks' sets", "frames_per_gpu": F,
                       "points_per_step": world * n, "parallelism": "frames sharded, %d rank(s), one process per GPU, "
                                                                    "r3d_comm (%s)" % (world, comm.rccl_origin())},
            "union_voxels": st["voxels"], "union_overflow": st["overflow"],
            "fabric_bytes_in_per_gpu": 8 * st["voxels"] * (world - 1) // max(world, 1),
            "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None, "kernel": "fuse_rgb_kernel<f32,pose>",
                         "kernel_ms": round(ms, 5), "algorithmic_bytes_per_launch": n * 23}}), flush=True)
    comm.barrier()
    comm.close()
    ctx.close()

