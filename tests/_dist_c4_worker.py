"""Worker for tests/test_gpu_eight_ranks.py: BASELINE config 4 WHOLE -- 1000 frames of 1280x384 (491.52 M points, 5.9 GB of
f32 xyz) through dist.ShardedFusion over four ranks that share the box's GPU (stand-in transport, tests/c/mock_rccl.cpp via
R3D_RCCL_PATH; gloo carries the communicator id only).  Every rank assembles the world cloud both ways ('outputs': fuse own
250 frames, all-gather xyz; 'inputs': all-gather rasters + poses, fuse all 1000 here) and compares EVERY BIT of it with the
cloud of one launch over all 1000 frames."""
import importlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
r3d = importlib.import_module("3d_reconstruction_system_amd")
D = importlib.import_module("3d_reconstruction_system_amd.dist")
CM = importlib.import_module("3d_reconstruction_system_amd.comm")


def main():
    out_path, F = sys.argv[1], int(sys.argv[2])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda", 0)
    ctx = r3d.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
    box = [CM.Comm.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    comm = CM.Comm(ctx, box[0], rank, world)
    H, W = 384, 1280
    per = H * W
    rng = np.random.default_rng(4)
    table_all = torch.from_numpy(r3d.pose_table(rng.normal(size=(F, 4)), rng.normal(size=(F, 3)) * 10)).to(dev)
    base = torch.from_numpy(rng.integers(1, 256, size=(8, H, W), dtype=np.uint8)).to(dev)      # 8 distinct rasters, repeated
    depth_all = base.repeat((F + 7) // 8, 1, 1)[:F].contiguous()
    cam = ctx.camera(H, W, *r3d.REF_INTRINSICS)
    want = torch.empty((F * per, 3), dtype=torch.float32, device=dev)
    r3d.fuse_frames_device(ctx, cam, depth_all.data_ptr(), np.uint8, F, table_all.data_ptr(), want.data_ptr(), np.float32)   # ONE launch
    lo, hi = D.shard_range(F, rank, world)
    counts = D.shard_counts(F, world)
    dd, pp = depth_all[lo:hi].contiguous(), table_all[lo:hi].contiguous()
    eng = D.ShardedFusion(H, W, r3d.REF_INTRINSICS, out_dtype="float32", transport=D.R3dTransport(comm, CM.GATHER_AUTO))
    notes, ok = [], True
    full = torch.full((F * per, 3), float("nan"), dtype=torch.float32, device=dev)
    for name, run in (("outputs", lambda: eng.fuse_and_gather(dd, pp, counts, out=full)),
                      ("inputs", lambda: eng.gather_inputs_and_fuse(dd, pp, counts, out=full))):
        full.fill_(float("nan"))
        got = run()
        torch.cuda.synchronize()
        same = bool(torch.equal(got.view(torch.int32), want.view(torch.int32)))          # bits, not values (NaN-proof)
        notes.append("%s: %s" % (name, same))
        ok = ok and same
    with open("%s.rank%d" % (out_path, rank), "w") as f:
        f.write("ok=%d lo=%d hi=%d points=%d bytes=%d origin=%s | %s\n" % (ok, lo, hi, F * per, F * per * 12, comm.rccl_origin(), "; ".join(notes)))
    dist.barrier()
    comm.close()
    ctx.close()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
