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


def all_gather_cloud(shard, points_per_rank, group=None):
    """All-gather xyz shards of possibly unequal length into the full cloud, rank order.
    shard: torch tensor [n_local, 3]; points_per_rank: list of n_local for every rank.
    Equal shards use all_gather_into_tensor straight into the result (no copy); unequal ones
    are padded to the longest shard and compacted."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    assert len(points_per_rank) == world and shard.shape[0] == points_per_rank[dist.get_rank(group)]
    longest = max(points_per_rank)
    total = sum(points_per_rank)
    if all(c == longest for c in points_per_rank):
        full = torch.empty((total, 3), dtype=shard.dtype, device=shard.device)
        dist.all_gather_into_tensor(full, shard.contiguous(), group=group)
        return full
    padded = torch.zeros((longest, 3), dtype=shard.dtype, device=shard.device)
    padded[:shard.shape[0]] = shard
    gathered = torch.empty((world * longest, 3), dtype=shard.dtype, device=shard.device)
    dist.all_gather_into_tensor(gathered, padded, group=group)
    parts = [gathered[r * longest:r * longest + c] for r, c in enumerate(points_per_rank) if c]
    return torch.cat(parts, dim=0) if parts else gathered[:0]


class ShardedFusion:
    """Per-rank engine: fuse this rank's frames on its GPU, then all-gather.

    fuse_fn(depth_tensor [F,H,W], pose_tensor [F,12] float64, out_tensor [F*H*W,3]) -> None
    defaults to the HIP kernel on the tensors' device (requires CUDA/ROCm tensors); CPU tests of
    the sharding / gather logic inject a checker function instead."""

    def __init__(self, height, width, intrinsics, out_dtype="float32", device=None, fuse_fn=None, group=None):
        import torch
        self.torch = torch
        self.h, self.w = int(height), int(width)
        self.intrinsics = tuple(intrinsics)
        self.out_dtype = getattr(torch, out_dtype) if isinstance(out_dtype, str) else out_dtype
        self.device = torch.device(device) if device is not None else torch.device("cpu")
        self.group = group
        self._ctx = None
        self._cam = None
        self.fuse_fn = fuse_fn or self._hip_fuse

    def _hip_fuse(self, depth, pose, out):
        torch = self.torch
        if depth.device.type != "cuda":
            raise RuntimeError("the HIP fusion path needs tensors on an MI355X (got %s); there is no CPU "
                               "fallback" % depth.device)
        from .device import Context
        from .fusion import fuse_frames_device
        if self._ctx is None:
            idx = depth.device.index if depth.device.index is not None else torch.cuda.current_device()
            self._ctx = Context(idx, stream=torch.cuda.current_stream(depth.device).cuda_stream)
            self._cam = self._ctx.camera(self.h, self.w, *self.intrinsics)
        np_depth = {torch.uint8: np.uint8, torch.uint16: np.uint16, torch.float32: np.float32}[depth.dtype]
        np_out = {torch.float32: np.float32, torch.float64: np.float64}[out.dtype]
        fuse_frames_device(self._ctx, self._cam, depth.data_ptr(), np_depth, depth.shape[0], pose.data_ptr(),
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
        """'outputs' assembly: fuse this rank's frames, all-gather the xyz shards."""
        shard = self.fuse_local(depth, pose, out)
        pts = [c * self.h * self.w for c in frames_per_rank]
        return all_gather_cloud(shard, pts, self.group)

    def gather_inputs(self, depth, pose, frames_per_rank):
        """All-gather the depth rasters and pose rows of every rank (padded to the longest shard).
        Returns (depth_all [world*Fmax,H,W], pose_all [world*Fmax,12], Fmax)."""
        import torch.distributed as dist
        torch = self.torch
        world = dist.get_world_size(self.group)
        fmax = max(frames_per_rank)
        if depth.shape[0] != fmax:
            pad_d = torch.zeros((fmax, self.h, self.w), dtype=depth.dtype, device=depth.device)
            pad_d[:depth.shape[0]] = depth
            pad_p = torch.zeros((fmax, 12), dtype=pose.dtype, device=pose.device)
            pad_p[:pose.shape[0]] = pose
            depth, pose = pad_d, pad_p
        depth_all = torch.empty((world * fmax, self.h, self.w), dtype=depth.dtype, device=depth.device)
        pose_all = torch.empty((world * fmax, 12), dtype=pose.dtype, device=pose.device)
        dist.all_gather_into_tensor(depth_all, depth.contiguous(), group=self.group)
        dist.all_gather_into_tensor(pose_all, pose.contiguous(), group=self.group)
        return depth_all, pose_all, fmax

    def gather_inputs_and_fuse(self, depth, pose, frames_per_rank, out=None):
        """'inputs' assembly: all-gather rasters + poses, fuse every frame here.  Same bits as
        fuse_and_gather (same kernel, same per-frame arithmetic), 1/12 of the fabric traffic."""
        torch = self.torch
        depth_all, pose_all, fmax = self.gather_inputs(depth, pose, frames_per_rank)
        total = sum(frames_per_rank)
        per = self.h * self.w
        if out is None:
            out = torch.empty((total * per, 3), dtype=self.out_dtype, device=depth.device)
        if all(c == fmax for c in frames_per_rank):
            if total:
                self.fuse_fn(depth_all, pose_all, out)          # one launch over world*F frames
            return out
        done = 0
        for r, c in enumerate(frames_per_rank):                  # ragged: skip each rank's padding
            if c:
                self.fuse_fn(depth_all[r * fmax:r * fmax + c], pose_all[r * fmax:r * fmax + c],
                             out[done * per:(done + c) * per])
            done += c
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
    merged = torch.cat(parts).cpu().numpy().view(np.uint64)
    return np.unique(merged)
