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
