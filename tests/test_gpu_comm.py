"""GPU: the exchange step behind the C ABI (r3d_comm_*: RCCL dlopen'ed by libr3d_hip.so itself).  The test box has ONE
GPU and RCCL refuses two ranks on one device, so what runs here is a world of one rank: library binding, communicator
set-up, every algorithm's degenerate case, and the sharded engine driven through R3dTransport.  The offset / ragged
logic for N > 1 is covered by the gloo tests (same ShardedFusion code, TorchTransport) and by tests/c/comm_2rank.c on
boxes with several GPUs."""
import importlib
import os
import subprocess

import numpy as np
import pytest

from helpers import PKG, ROOT, r3d as _r3d
from oracle import fusion_ref as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def R():
    return _r3d()


def test_comm_single_rank_allgather_allreduce(R, real_rccl):
    CM = importlib.import_module(PKG + ".comm")
    ctx = R.Context(0)
    comm = CM.Comm(ctx, CM.Comm.unique_id(), 0, 1)
    assert comm.rccl_origin() != ""
    rng = np.random.default_rng(0)
    src = rng.integers(0, 256, 100003, dtype=np.uint8)
    d_src, d_dst = ctx.alloc(src.nbytes).upload(src), ctx.alloc(src.nbytes)
    for algo in (CM.GATHER_AUTO, CM.GATHER_NCCL, CM.GATHER_DIRECT):
        R.load_library().r3d_memset(ctx.handle, d_dst.ptr, 0, src.nbytes)
        comm.allgather(d_src.ptr, [src.nbytes], d_dst.ptr, algo)
        np.testing.assert_array_equal(d_dst.download(np.uint8, src.nbytes), src)
    comm.allgather(d_dst.ptr, [src.nbytes], d_dst.ptr, CM.GATHER_DIRECT)          # in place
    np.testing.assert_array_equal(d_dst.download(np.uint8, src.nbytes), src)
    sums = rng.normal(size=18)
    d_s = ctx.alloc(18 * 8).upload(sums)
    comm.allreduce_sum_f64(d_s.ptr, 18)
    np.testing.assert_array_equal(d_s.download(np.float64, 18), sums)
    with pytest.raises(ValueError):
        comm.allgather(d_src.ptr, [1, 2], d_dst.ptr)
    with pytest.raises(R.R3DError):
        CM.Comm(ctx, CM.Comm.unique_id(), 3, 2)
    comm.close()
    ctx.close()


def test_sharded_engine_over_r3d_transport_single_rank(R, real_rccl):
    import torch
    CM = importlib.import_module(PKG + ".comm")
    D = importlib.import_module(PKG + ".dist")
    dev = torch.device("cuda", 0)
    ctx = R.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
    comm = CM.Comm(ctx, CM.Comm.unique_id(), 0, 1)
    rng = np.random.default_rng(4)
    F, H, W = 5, 48, 64
    depth = rng.integers(0, 256, size=(F, H, W), dtype=np.uint8)
    q, t = rng.normal(size=(F, 4)), rng.normal(size=(F, 3)) * 10
    table = R.pose_table(q, t)
    eng = D.ShardedFusion(H, W, R.REF_INTRINSICS, out_dtype="float32", transport=D.R3dTransport(comm))
    dd, pp = torch.from_numpy(depth).to(dev), torch.from_numpy(table).to(dev)
    a = eng.fuse_and_gather(dd, pp, [F])
    b = eng.gather_inputs_and_fuse(dd, pp, [F])
    out = torch.empty((F * H * W, 3), dtype=torch.float32, device=dev)
    c = eng.fuse_and_gather(dd, pp, [F], out=out)
    torch.cuda.synchronize()
    want = R.fuse_frames(depth, q, t, ctx=ctx)
    for got in (a, b, c):
        np.testing.assert_array_equal(got.cpu().numpy(), want)
    e_norm, e_comp = O.parity_errors(want, O.fuse_frames(depth, q, t))
    assert e_norm <= 1e-6 and e_comp <= 1e-4
    # the engine follows the caller's stream: a side stream gets its own context
    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        d = eng.fuse_local(dd, pp)
    s.synchronize()
    np.testing.assert_array_equal(d.cpu().numpy(), want)
    assert len(eng._ctxs) == 2
    # ... and the exchange orders itself with that stream: compute on the side stream, the collective on the comm's
    # stream.  Many rounds with a busy side stream: a missing wait would ship the shard before the fuse wrote it.
    with torch.cuda.stream(s):
        for k in range(12):
            big = torch.empty((F * H * W, 3), dtype=torch.float32, device=dev)
            filler = torch.randn(1 << 22, device=dev).sin_().sum()         # keeps the side stream behind the host
            got = eng.fuse_and_gather(dd, pp, [F], out=big)
            gi = eng.gather_inputs_and_fuse(dd, pp, [F])
            assert torch.equal(got, torch.from_numpy(want).to(dev)) and torch.equal(gi, got), k
            del filler
    s.synchronize()
    comm.close()
    ctx.close()


def test_plain_c_multi_rank_consumer(tmp_path, real_rccl):
    exe = str(tmp_path / "comm_2rank")
    libdir = os.path.join(ROOT, PKG)
    build = subprocess.run(["gcc", "-std=gnu99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                            os.path.join(ROOT, "tests", "c", "comm_2rank.c"), "-o", exe, "-L", libdir, "-lr3d_hip", "-lm",
                            "-Wl,-rpath," + libdir], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr
    one = subprocess.run([exe, "1", "5"], capture_output=True, text=True, timeout=120)        # a world of one rank
    assert one.returncode == 0 and "identical to the single-GPU cloud" in one.stdout, one.stdout + one.stderr
    two = subprocess.run([exe, "2", "5"], capture_output=True, text=True, timeout=120)        # runs only with >= 2 GPUs
    assert two.returncode in (0, 77), two.stdout + two.stderr


# ---- N > 1 on the one-GPU box: the C ABI's exchange logic against a stand-in transport ----------------------------------
@pytest.mark.parametrize("world,n_frames,algo", [(2, 5, 0), (2, 6, 1), (3, 7, 2), (4, 9, 0), (4, 8, 1), (3, 2, 0)])
def test_plain_c_consumer_multi_rank_over_mock_transport(tmp_path, mock_rccl, world, n_frames, algo):
    """world ranks share the GPU; ragged blocks (and an EMPTY last block: 3 ranks, 2 frames), both assemblies, the
    all-reduce; algo 0 auto / 1 ncclAllGather (equal shards) / 2 direct send/recv."""
    exe = str(tmp_path / "comm_2rank")
    libdir = os.path.join(ROOT, PKG)
    build = subprocess.run(["gcc", "-std=gnu99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                            os.path.join(ROOT, "tests", "c", "comm_2rank.c"), "-o", exe, "-L", libdir, "-lr3d_hip", "-lm",
                            "-Wl,-rpath," + libdir], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr
    run = subprocess.run([exe, str(world), str(n_frames), str(algo)], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, R3D_RCCL_PATH=mock_rccl, R3D_SHARE_GPU="1"))
    assert run.returncode == 0, run.stdout + run.stderr
    assert run.stdout.count("identical to the single-GPU cloud") == world, run.stdout


@pytest.mark.parametrize("world,n_frames", [(2, 6), (2, 5), (3, 7)])
def test_sharded_engine_multi_rank_over_mock_transport(tmp_path, mock_rccl, world, n_frames):
    out = str(tmp_path / "res")
    port = 29800 + (os.getpid() + 7 * n_frames + world) % 150
    import sys
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world, "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "_dist_mock_worker.py"), out, str(n_frames)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", R3D_RCCL_PATH=mock_rccl))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    for rank in range(world):
        line = open("%s.rank%d" % (out, rank)).read()
        assert "ok=1" in line and "R3D_RCCL_PATH" in line, line
