"""CPU: the N>1 path (frame sharding + all-gather) with world_size 2 over gloo."""
import importlib
import os
import subprocess
import sys

import pytest

from helpers import PKG, ROOT


def test_shard_ranges_cover_frames_in_order():
    D = importlib.import_module(PKG + ".dist")
    for n in (0, 1, 2, 7, 8, 9, 100, 1000, 1001):
        for world in (1, 2, 3, 4, 8):
            prev = 0
            for r in range(world):
                lo, hi = D.shard_range(n, r, world)
                assert lo == prev and lo <= hi <= n
                prev = hi
            assert prev == n
            assert sum(D.shard_counts(n, world)) == n
            assert max(D.shard_counts(n, world)) == -(-n // world) if n else True


@pytest.mark.parametrize("n_frames", [6, 5, 1])   # equal shards, ragged tail, one rank empty
def test_two_rank_fuse_and_gather(tmp_path, n_frames):
    out = str(tmp_path / "res")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    port = 29500 + (os.getpid() + n_frames) % 2000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "_dist_worker.py"), out, str(n_frames)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    for rank in range(2):
        line = open("%s.rank%d" % (out, rank)).read()
        assert "ok=1" in line and "refused=1" in line, line


def test_eight_ranks_ragged_1001_frames_over_gloo(tmp_path):
    """The shard / pad / gather logic of dist.py at EIGHT ranks on a ragged 1001-frame job (7 x 126 + 119): the GPU boxes of
    the suite allow six processes on the card, so eight ranks are rehearsed here, on the CPU, with the oracle as the compute
    hook (tests/_dist_worker.py) -- both assemblies and the map union against the single-process answer."""
    out = str(tmp_path / "res")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    port = 29500 + (os.getpid() + 1001) % 2000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "_dist_worker.py"), out, "1001"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    for rank in range(8):
        line = open("%s.rank%d" % (out, rank)).read()
        assert "ok=1" in line and "refused=1" in line, line
    assert "lo=0 hi=126" in open(out + ".rank0").read() and "lo=882 hi=1001" in open(out + ".rank7").read()


def test_byte_offsets_of_config_4_and_5_over_eight_ranks():
    """Where every rank's shard starts in the gathered world cloud, in BYTES, at the sizes the 8-GPU configs have: config 4's
    125 frames of 1280x384 per rank (737 MB each: the sixth rank's slot already starts beyond 2^32), config 5's 250 frames of
    1920x1080 (6.2 GB each).  The C ABI takes and keeps these as int64 (include/r3d.h: r3d_comm_allgather)."""
    D = importlib.import_module(PKG + ".dist")
    for frames, hw, world in ((1000, 384 * 1280, 8), (2000, 1080 * 1920, 8), (1001, 384 * 1280, 8)):
        counts = D.shard_counts(frames, world)
        byte_counts = [c * hw * 12 for c in counts]
        offs = [sum(byte_counts[:r]) for r in range(world + 1)]
        assert offs[-1] == frames * hw * 12 and all(b >= 0 for b in byte_counts)
        assert offs[-1] > 1 << 32 and any(o > 1 << 32 for o in offs[:-1])
        import numpy as np
        as_i64 = np.asarray(byte_counts, dtype=np.int64)                      # what comm.Comm._counts hands to the C ABI
        assert int(as_i64.sum()) == offs[-1] and as_i64.dtype.itemsize == 8
