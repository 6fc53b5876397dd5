"""Frame-sharded fusion across the GPUs of one node: one process per GPU, frames split into
contiguous blocks in pose-file order (the frame loop of camera_to_world.py:149-172 carries no
state between frames).  torch.distributed is the transport (backend "nccl" = RCCL over xGMI on
the GPU box, "gloo" in CPU tests); the per-point arithmetic is the HIP library's.

Assembling the fused world cloud on every rank, two ways with bit-identical results:
  * "outputs": every rank fuses its own frames, then ONE all-gather of the xyz shards
    (12 B/point over xGMI).
  * "inputs":  ONE all-gather of the depth rasters + pose rows (1 B/point + 96 B/frame over xGMI),
    then every rank fuses ALL frames locally.  On MI355X the fused kernel streams 6.6 TB/s of
    HBM while an xGMI link moves ~0.1-0.15 TB/s, so re-computing 13 B/point of HBM traffic is
    far cheaper than receiving 12 B/point over the fabric: this is the default.
"""
import numpy as np


def shard_range(n_frames, rank, world):
    """Contiguous block [lo, hi) of frames for `rank`: ceil(n/world) per rank, the tail ranks
    may get fewer (or none).  Concatenating the blocks in rank order restores frame order."""
    per = -(-n_frames // world) if world > 0 else n_frames
    lo = min(rank * per, n_frames)
    hi = min(lo + per, n_frames)
    return lo, hi


def shard_counts(n_frames, world):
    return [shard_range(n_frames, r, world)[1] - shard_range(n_frames, r, world)[0] for r in range(world)]


class TorchTransport:
    """The exchange step over torch.distributed ("nccl" = RCCL on the GPU box, "gloo" in CPU tests)."""

    name = "torch.distributed"

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)

    def allgather_rows(self, shard, rows_per_rank, out=None):
        """Shards [rows_r, ...] of possibly unequal length -> [sum rows, ...] in rank order.  Equal shards go straight
        into the result (all_gather_into_tensor, no copy); unequal ones are padded to the longest and compacted."""
        import torch
        assert len(rows_per_rank) == self.world and shard.shape[0] == rows_per_rank[self.rank]
        tail = tuple(shard.shape[1:])
        longest, total = max(rows_per_rank), sum(rows_per_rank)
        if out is None:
            out = torch.empty((total,) + tail, dtype=shard.dtype, device=shard.device)
        if total == 0:
            return out
        if all(c == longest for c in rows_per_rank):
            self.dist.all_gather_into_tensor(out, shard.contiguous(), group=self.group)
            return out
        padded = torch.zeros((longest,) + tail, dtype=shard.dtype, device=shard.device)
        padded[:shard.shape[0]] = shard
        gathered = torch.empty((self.world * longest,) + tail, dtype=shard.dtype, device=shard.device)
        self.dist.all_gather_into_tensor(gathered, padded, group=self.group)
        done = 0
        for r, c in enumerate(rows_per_rank):
            if c:
                out[done:done + c] = gathered[r * longest:r * longest + c]
            done += c
        return out


class R3dTransport:
    """The exchange step through the C ABI (comm.Comm: RCCL bound by libr3d_hip.so, unequal shards by grouped
    send/recv, no padding, no torch.distributed).  Tensors only lend their HBM addresses; the transfers run on the
    comm's ctx stream.  When that is not the torch stream current at the call (e.g. under `with torch.cuda.stream(s)`), the
    two are ordered here: the comm stream waits for what torch has enqueued so far (the fuse that wrote the shard), torch's
    stream waits for the collective, and both tensors are recorded on the comm stream so that the caching allocator does not
    hand their memory out while it is still in flight there."""

    name = "r3d_comm"

    def __init__(self, comm, algo=0):
        self.comm, self.algo = comm, int(algo)
        self.rank, self.world = comm.rank, comm.world

    def _comm_stream(self, device):
        """The comm ctx's raw hipStream_t as a torch stream object, or None when it IS torch's current stream."""
        import torch
        handle = self.comm.ctx.stream_handle()
        cur = torch.cuda.current_stream(device)
        if handle == cur.cuda_stream:
            return None, cur
        ext = torch.cuda.default_stream(device) if handle == 0 else torch.cuda.ExternalStream(handle, device=device)
        return ext, cur

    def allgather_rows(self, shard, rows_per_rank, out=None):
        import torch
        assert len(rows_per_rank) == self.world and shard.shape[0] == rows_per_rank[self.rank]
        if shard.device.type != "cuda":
            raise RuntimeError("r3d_comm moves HBM buffers; got a tensor on %s" % shard.device)
        idx = shard.device.index if shard.device.index is not None else torch.cuda.current_device()
        if idx != self.comm.ctx.device:
            raise RuntimeError("the communicator lives on GPU %d, the shard on GPU %d" % (self.comm.ctx.device, idx))
        tail = tuple(shard.shape[1:])
        row_bytes = shard.element_size()
        for d in tail:
            row_bytes *= d
        if out is None:
            out = torch.empty((sum(rows_per_rank),) + tail, dtype=shard.dtype, device=shard.device)
        shard = shard.contiguous()
        ext, cur = self._comm_stream(shard.device)
        if ext is not None:
            ext.wait_stream(cur)                # producers of `shard` (and earlier readers of `out`) first
        self.comm.allgather(shard.data_ptr(), [c * row_bytes for c in rows_per_rank], out.data_ptr(), self.algo)
        if ext is not None:
            cur.wait_stream(ext)                # consumers on torch's stream see the gathered rows
            shard.record_stream(ext)
            out.record_stream(ext)
        return out


def all_gather_cloud(shard, points_per_rank, group=None, transport=None):
    """All-gather xyz shards of possibly unequal length into the full cloud, rank order.
    shard: torch tensor [n_local, 3]; points_per_rank: list of n_local for every rank."""
    return (transport or TorchTransport(group)).allgather_rows(shard, list(points_per_rank))


class ShardedFusion:
    """Per-rank engine: fuse this rank's frames on its GPU, then all-gather.

    fuse_fn(depth_tensor [F,H,W], pose_tensor [F,12] float64, out_tensor [F*H*W,3]) -> None
    defaults to the HIP kernel on the tensors' device (requires CUDA/ROCm tensors); CPU tests of
    the sharding / gather logic inject a checker function instead.
    transport: TorchTransport (default) or R3dTransport."""

    def __init__(self, height, width, intrinsics, out_dtype="float32", device=None, fuse_fn=None, group=None,
                 transport=None):
        import torch
        self.torch = torch
        self.h, self.w = int(height), int(width)
        self.intrinsics = tuple(intrinsics)
        self.out_dtype = getattr(torch, out_dtype) if isinstance(out_dtype, str) else out_dtype
        self.device = torch.device(device) if device is not None else torch.device("cpu")
        self.group = group
        self._transport = transport
        self._ctxs = {}
        self.fuse_fn = fuse_fn or self._hip_fuse

    @property
    def transport(self):
        if self._transport is None:
            self._transport = TorchTransport(self.group)
        return self._transport

    def _ctx_for(self, device):
        """(Context, Camera) bound to the torch stream that is current NOW on `device`: a context launches on the
        stream it was created with, so it is cached per (device, stream) -- a caller that switches streams
        (torch.cuda.stream(s)) gets a context on that stream, ordered with its producers and consumers.  The comm's own
        context is reused when it sits on that very stream; otherwise R3dTransport.allgather_rows orders the two streams."""
        torch = self.torch
        idx = device.index if device.index is not None else torch.cuda.current_device()
        stream = torch.cuda.current_stream(device).cuda_stream
        key = (idx, stream)
        hit = self._ctxs.get(key)
        if hit is None:
            from .device import Context
            c = None
            if isinstance(self._transport, R3dTransport) and self._transport.comm.ctx.device == idx \
                    and self._transport.comm.ctx.stream_handle() == stream:
                c = self._transport.comm.ctx          # compute and exchange share one stream
            if c is None:
                c = Context(idx, stream=stream)
            hit = (c, c.camera(self.h, self.w, *self.intrinsics))
            self._ctxs[key] = hit
        return hit

    def _hip_fuse(self, depth, pose, out):
        torch = self.torch
        if depth.device.type != "cuda":
            raise RuntimeError("the HIP fusion path needs tensors on an MI355X (got %s); there is no CPU "
                               "fallback" % depth.device)
        from .fusion import fuse_frames_device
        ctx, cam = self._ctx_for(depth.device)
        np_depth = {torch.uint8: np.uint8, torch.uint16: np.uint16, torch.float32: np.float32}[depth.dtype]
        np_out = {torch.float32: np.float32, torch.float64: np.float64}[out.dtype]
        fuse_frames_device(ctx, cam, depth.data_ptr(), np_depth, depth.shape[0], pose.data_ptr(),
                           out.data_ptr(), np_out)

    def fuse_local(self, depth, pose, out=None):
        torch = self.torch
        f = depth.shape[0]
        assert tuple(depth.shape[1:]) == (self.h, self.w) and tuple(pose.shape) == (f, 12)
        if out is None:
            out = torch.empty((f * self.h * self.w, 3), dtype=self.out_dtype, device=depth.device)
        if f:
            self.fuse_fn(depth.contiguous(), pose.contiguous(), out)
        return out

    def fuse_and_gather(self, depth, pose, frames_per_rank, out=None):
        """'outputs' assembly: fuse this rank's frames, all-gather the xyz shards (12 B/point over xGMI).
        out: optional preallocated [total points, 3] tensor; this rank then fuses straight into its slot."""
        per = self.h * self.w
        pts = [c * per for c in frames_per_rank]
        t = self.transport
        if out is not None:
            lo = sum(pts[:t.rank])
            shard = self.fuse_local(depth, pose, out[lo:lo + pts[t.rank]])
        else:
            shard = self.fuse_local(depth, pose)
        return t.allgather_rows(shard, pts, out)

    def gather_inputs(self, depth, pose, frames_per_rank):
        """All-gather the depth rasters and pose rows of every rank, compact, in rank order:
        (depth_all [total F,H,W], pose_all [total F,12])."""
        t = self.transport
        return (t.allgather_rows(depth.contiguous(), list(frames_per_rank)),
                t.allgather_rows(pose.contiguous(), list(frames_per_rank)))

    def gather_inputs_and_fuse(self, depth, pose, frames_per_rank, out=None):
        """'inputs' assembly: all-gather rasters + poses (1 B/point + 96 B/frame), fuse every frame here in ONE
        launch.  Same bits as fuse_and_gather (same kernel, same per-frame arithmetic), 1/12 of the fabric traffic."""
        torch = self.torch
        depth_all, pose_all = self.gather_inputs(depth, pose, frames_per_rank)
        if isinstance(self.transport, TorchTransport) and self.fuse_fn == self._hip_fuse and depth.device.type == "cuda":
            self._ctx_for(depth.device)[0].inputs_fresh()      # torch's collective wrote them: not in the cache, not tracked
        total = sum(frames_per_rank)
        if out is None:
            out = torch.empty((total * self.h * self.w, 3), dtype=self.out_dtype, device=depth.device)
        if total:
            self.fuse_fn(depth_all, pose_all, out)
        return out


def all_gather_voxel_codes(codes, group=None):
    """Union of every rank's occupied-voxel set (config 5: frames sharded, ONE map).  codes: this rank's distinct Morton
    codes (uint64 NumPy array, any order).  Each rank inserts its own shard of the world cloud into its own HBM hash
    set (voxelmap.VoxelSet) -- 12 B/point never leave the GPU -- and only the distinct codes (8 B/voxel) cross the
    fabric: all-gather of padded int64 tensors, then sort + unique.  Returns the ascending union on every rank."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    mine = torch.from_numpy(np.ascontiguousarray(codes, dtype=np.uint64).view(np.int64)).to(dev)
    n_local = torch.tensor([mine.numel()], dtype=torch.int64, device=dev)
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, n_local, group=group)
    counts = [int(c.item()) for c in counts]
    longest = max(counts) if counts else 0
    if longest == 0:
        return np.zeros(0, dtype=np.uint64)
    padded = torch.zeros(longest, dtype=torch.int64, device=dev)
    padded[:mine.numel()] = mine
    gathered = torch.empty(world * longest, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(gathered, padded, group=group)
    parts = [gathered[r * longest:r * longest + c] for r, c in enumerate(counts) if c]
    merged = torch.cat(parts)
    if merged.device.type == "cuda":
        # sort in HBM with the library's radix sort (48-bit Morton codes), drop repeats, then one D2H of the union
        from . import _lib as L
        from .device import Context
        ctx = Context(merged.device.index, stream=torch.cuda.current_stream(merged.device).cuda_stream)
        try:
            L.check(ctx.lib.r3d_sort_u64(ctx.handle, merged.data_ptr(), merged.numel(), 48))
            merged = torch.unique_consecutive(merged)
            torch.cuda.current_stream(merged.device).synchronize()
        finally:
            ctx.close()
        return merged.cpu().numpy().view(np.uint64)
    return np.unique(merged.numpy().view(np.uint64))


def fuse_pose_file_sharded(qt_path, depth_dir, intrinsics, out_dtype, ctx, comm, everywhere=False, algo=0, pose_scale=1.0, keep_on_device=False):
    """BASELINE config 4 as the drop-in runs it: the pose file's frames in contiguous blocks, one block per rank
    (camera_to_world.py:149-172 carries no state between frames).  Every rank decodes ITS depth PNGs, fuses them with one
    launch straight into its slot of the world cloud in HBM, and one all-gather over the C ABI (r3d_allgather_xyz: RCCL,
    unequal blocks) assembles the cloud on every GPU.  No torch.

    Returns (names, lo, hi, depths_local [hi-lo,H,W], world): `world` is the whole [F*H*W,3] cloud on rank 0 (on every rank
    with everywhere=True), else None.  Same bits as the single-GPU fuse_frames: same kernel, same per-frame arithmetic.
    keep_on_device=True: nothing is downloaded; `world` is a dict instead -- {"d_full": DeviceBuffer holding the whole cloud on
    THIS rank's GPU, "d_depth": DeviceBuffer with this rank's rasters (None for an empty block), "per": points per frame} --
    whose buffers the caller frees (the drop-in formats its text from them on the GPU, device_text.TextWriter)."""
    from . import cloud_io
    from .device import xyz_code
    from .fusion import fuse_frames_device
    from .poses import pose_table, read_pose_file
    import os
    names, quats, ts = read_pose_file(qt_path)
    n_frames = len(names)
    rank, world_size = comm.rank, comm.world
    lo, hi = shard_range(n_frames, rank, world_size)
    out_dtype = np.dtype(out_dtype)
    xyz_code(out_dtype)
    if n_frames == 0:
        if keep_on_device:
            return names, 0, 0, np.empty((0, 0, 0), np.uint8), {"d_full": None, "d_depth": None, "per": 0}
        return names, 0, 0, np.empty((0, 0, 0), np.uint8), (np.empty((0, 3), out_dtype) if rank == 0 or everywhere else None)
    paths = [os.path.join(depth_dir, n) for n in names]
    # a rank whose block is empty (more GPUs than frames) still needs the raster size: the first frame's.
    # Every rank must enter the collective below or none: a rank-local failure (missing / corrupt PNG) or rasters that differ
    # between blocks would leave the others waiting in RCCL for ever (it has no timeout).  So the ranks first agree on
    # (ok, H, W, bytes per depth value) with one small all-gather and raise the SAME error everywhere.
    err, probe = None, None
    try:
        probe = cloud_io.read_depth_batch(paths[lo:hi] if hi > lo else paths[:1])
    except Exception as e:          # noqa: BLE001 -- reported on every rank below
        err = e
    mine_row = np.array([0.0 if err is None else 1.0, 0.0 if probe is None else probe.shape[1], 0.0 if probe is None else probe.shape[2],
                         0.0 if probe is None else probe.dtype.itemsize], dtype=np.float64)
    d_rows = ctx.alloc(8 * 4 * (world_size + 1))
    try:
        from . import _lib as L
        L.check(ctx.lib.r3d_memcpy_h2d(ctx.handle, d_rows.ptr + 32 * world_size, mine_row.ctypes.data, 32))
        comm.allgather(d_rows.ptr + 32 * world_size, [32] * world_size, d_rows.ptr)
        rows = d_rows.download(np.float64, 4 * world_size).reshape(world_size, 4)
    finally:
        d_rows.free()
    bad = [r for r in range(world_size) if rows[r, 0] != 0]
    if bad:
        raise RuntimeError("rank(s) %s could not read their depth frames%s" % (bad, "" if err is None else ": %s" % err))
    if not (np.all(rows[:, 1:] == rows[0, 1:])):
        raise ValueError("depth rasters differ between the ranks' frame blocks (H, W, bytes): %s" % rows[:, 1:].tolist())
    depths = probe if hi > lo else probe[:0]
    h, w = probe.shape[1], probe.shape[2]
    per = h * w
    row = 3 * out_dtype.itemsize
    counts = shard_counts(n_frames, world_size)
    pts = [c * per for c in counts]
    d_full = ctx.alloc(n_frames * per * row)
    bufs = [d_full]
    d_depth = None
    try:
        mine = d_full.ptr + lo * per * row
        if hi > lo:
            cam = ctx.camera(h, w, *intrinsics)
            d_depth = ctx.alloc(depths.nbytes).upload(depths)
            table = pose_table(quats[lo:hi], ts[lo:hi] * float(pose_scale))
            d_pose = ctx.alloc(table.nbytes).upload(table)
            bufs += [d_depth, d_pose]
            fuse_frames_device(ctx, cam, d_depth.ptr, depths.dtype, hi - lo, d_pose.ptr, mine, out_dtype)
        comm.allgather_xyz(mine, pts, out_dtype, d_full.ptr, algo)
        world = None
        if keep_on_device:
            ctx.sync()
            world = {"d_full": d_full, "d_depth": d_depth, "per": per}
            bufs = [b for b in bufs if b is not d_full and b is not d_depth]
        elif rank == 0 or everywhere:
            world = d_full.download(out_dtype, n_frames * per * 3).reshape(-1, 3)
        else:
            ctx.sync()
    finally:
        for b in bufs:
            b.free()
    return names, lo, hi, depths, world
