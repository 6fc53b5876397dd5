"""Worker for tests/test_gpu_dist.py: 2 ranks share the one GPU of the test box (gloo moves the bytes;
on the real node the same code runs one rank per GPU over RCCL).  Every rank fuses its frame block with
the HIP kernel through dist.ShardedFusion and the gathered cloud is checked against the oracle."""
import importlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import fusion_ref as O  # noqa: E402

D = importlib.import_module("3d_reconstruction_system_amd.dist")
P = importlib.import_module("3d_reconstruction_system_amd.poses")


def main():
    out_path, n_frames = sys.argv[1], int(sys.argv[2])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda", 0)
    H, W = 48, 64
    rng = np.random.default_rng(5)
    depths = rng.integers(0, 256, size=(n_frames, H, W), dtype=np.uint8)
    quats = rng.normal(size=(n_frames, 4))
    ts = rng.normal(size=(n_frames, 3)) * 10
    lo, hi = D.shard_range(n_frames, rank, world)
    counts = D.shard_counts(n_frames, world)
    eng = D.ShardedFusion(H, W, (O.REF_FX, O.REF_FY, O.REF_CX, O.REF_CY), out_dtype="float32")
    table = P.pose_table(quats[lo:hi], ts[lo:hi]) if hi > lo else np.zeros((0, 12))
    full = eng.fuse_and_gather(torch.from_numpy(depths[lo:hi].copy()).to(dev), torch.from_numpy(table).to(dev), counts)
    torch.cuda.synchronize()
    got = full.cpu().numpy()
    want = O.fuse_frames(depths, quats, ts)
    e_norm, e_comp = O.parity_errors(got, want)
    ok = got.shape == want.shape and e_norm <= 1e-6 and e_comp <= 1e-4
    full2 = eng.gather_inputs_and_fuse(torch.from_numpy(depths[lo:hi].copy()).to(dev), torch.from_numpy(table).to(dev),
                                       counts)
    torch.cuda.synchronize()
    ok = ok and np.array_equal(full2.cpu().numpy(), got)      # 'inputs' assembly: bit-identical
    with open("%s.rank%d" % (out_path, rank), "w") as f:
        f.write("ok=%d e_norm=%.3e lo=%d hi=%d\n" % (ok, e_norm, lo, hi))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
