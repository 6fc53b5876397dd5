"""r3d_comm: the multi-GPU exchange step through the C ABI (RCCL bound by libr3d_hip.so itself, no torch needed).

    id = Comm.unique_id()            # rank 0; ship the 128 bytes to every rank (file, pipe, MPI, torch broadcast ...)
    comm = Comm(ctx, id, rank, world)
    comm.allgather(d_send_ptr, byte_counts, d_recv_ptr)      # unequal shards, rank order, async on ctx's stream

One process per GPU (RCCL refuses two ranks on one device).

Launched by `python -m torch.distributed.run --nproc-per-node N script.py` (or any launcher that sets RANK, WORLD_SIZE,
LOCAL_RANK and, for several jobs on one node, MASTER_PORT), `Comm.from_env(ctx)` does the id exchange itself over an
abstract Unix socket -- one node, no torch, no files left behind.
"""
import atexit
import ctypes as C
import os
import socket
import time
import weakref

import numpy as np

from . import _lib as L

ID_BYTES = 128
GATHER_AUTO, GATHER_NCCL, GATHER_DIRECT = 0, 1, 2


def env_rank_world():
    """(rank, world) from the launcher's environment; (0, 1) when there is none."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if not 0 <= rank < world:
        raise ValueError("RANK=%d outside WORLD_SIZE=%d" % (rank, world))
    return rank, world


def env_local_device():
    """GPU index for this rank: LOCAL_RANK (one process per GPU).  R3D_SHARE_GPU=1 folds the ranks onto the visible GPUs
    (rehearsals on a one-GPU box against a stand-in transport; RCCL itself refuses two ranks on one device)."""
    local = int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0")))
    if os.environ.get("R3D_SHARE_GPU", "0") not in ("", "0"):
        n = C.c_int(0)
        L.check(L.load().r3d_device_count(C.byref(n)))
        return local % max(n.value, 1)
    return local


def _rendezvous_tag():
    return os.environ.get("R3D_RENDEZVOUS", "%s_%s_%s_%d" % (os.environ.get("MASTER_ADDR", "local"), os.environ.get("MASTER_PORT", "29500"),
                                                             os.environ.get("TORCHELASTIC_RUN_ID", "none"), os.getuid()))


def _rendezvous_name():
    # abstract-namespace Unix socket: lives only as long as rank 0 holds it, nothing to clean up or to go stale.
    # The name carries a digest of (MASTER_ADDR, MASTER_PORT, run id, uid): concurrent jobs on one node already need distinct
    # MASTER_PORTs for their launcher, and another user's job cannot collide with this one by accident.
    import hashlib
    return b"\0r3d_comm_" + hashlib.sha256(_rendezvous_tag().encode()).hexdigest()[:32].encode()


def _nonce():
    """16 bytes both sides derive from the job's tag: a stray or foreign connection that does not send them is ignored."""
    import hashlib
    return hashlib.sha256(b"r3d-nonce:" + _rendezvous_tag().encode()).digest()[:16]


def _recv_exact(conn, n):
    buf = b""
    while len(buf) < n:
        part = conn.recv(n - len(buf))
        if not part:
            return None
        buf += part
    return buf


def exchange_unique_id(rank, world, timeout=120.0):
    """The 128-byte communicator id on every rank of a ONE-NODE job: rank 0 creates it and serves it to the other
    world-1 ranks over an abstract Unix socket; they connect (retrying until rank 0 is up) and read it."""
    if world == 1:
        return Comm.unique_id()
    name = _rendezvous_name()
    deadline = time.monotonic() + timeout
    if rank == 0:
        uid = Comm.unique_id()
        srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        try:
            srv.bind(name)              # EADDRINUSE = another job with the same MASTER_PORT / R3D_RENDEZVOUS: fail loudly
            srv.listen(world)
            handed, confirmed = set(), set()      # got all 128 bytes from us / said so
            grace_until = None
            while True:
                if len(handed) == world - 1:
                    if confirmed == handed:
                        break
                    # Every rank has been handed the id but not every one has confirmed it: a rank whose read failed will
                    # knock again, so the name stays up for a short grace period (re-serving repeat hellos) instead of
                    # vanishing under it -- and a confirmation that never comes must not keep rank 0 here for long either.
                    if grace_until is None:
                        grace_until = time.monotonic() + 3.0
                    if time.monotonic() >= grace_until:
                        break
                wait_until = deadline if grace_until is None else grace_until
                srv.settimeout(max(wait_until - time.monotonic(), 0.01))
                try:
                    conn, _ = srv.accept()
                except socket.timeout:
                    if grace_until is not None:
                        break
                    raise TimeoutError("rendezvous: %d of %d ranks asked for the communicator id within %.0f s"
                                       % (len(handed), world - 1, timeout))
                with conn:
                    conn.settimeout(10.0)
                    try:
                        hello = _recv_exact(conn, 20)         # nonce + rank, exactly
                        if hello is None or hello[:16] != _nonce():
                            continue                          # not one of this job's ranks
                        who = int.from_bytes(hello[16:], "little")
                        if not 0 < who < world:
                            continue
                        conn.sendall(uid)
                        handed.add(who)
                        conn.settimeout(2.0)                  # the confirmation: one byte, waited for briefly
                        if _recv_exact(conn, 1) is not None:
                            confirmed.add(who)
                    except OSError:
                        pass                 # that rank will retry (or its confirmation was lost: grace period above)
        finally:
            srv.close()
        return uid
    while True:
        c = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        try:
            c.settimeout(10.0)
            c.connect(name)
            c.sendall(_nonce() + int(rank).to_bytes(4, "little"))
            uid = _recv_exact(c, ID_BYTES)
            if uid is not None:
                try:
                    c.sendall(b"\x01")      # confirmation (best effort: rank 0 does not depend on it)
                except OSError:
                    pass
                return uid
        except OSError:
            pass
        finally:
            c.close()
        if time.monotonic() > deadline:
            raise TimeoutError("rendezvous: rank 0 did not serve the communicator id within %.0f s" % timeout)
        time.sleep(0.05)


class Comm:
    def __init__(self, ctx, unique_id, rank, world):
        self.ctx = ctx
        self.rank, self.world = int(rank), int(world)
        uid = bytes(unique_id)
        if len(uid) != ID_BYTES:
            raise ValueError("unique id must be %d bytes" % ID_BYTES)
        buf = C.create_string_buffer(uid, ID_BYTES)
        h = C.c_void_p()
        L.check(ctx.lib.r3d_comm_create(ctx.handle, buf, self.rank, self.world, C.byref(h)))
        self.handle = h.value
        ctx.adopt(self)
        # RCCL must let go of the GPU before the interpreter starts tearing the HIP runtime down
        ref = weakref.ref(self)
        atexit.register(lambda: ref() is not None and ref().close())

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(ID_BYTES)
        L.check(L.load().r3d_comm_unique_id(buf))
        return buf.raw

    @classmethod
    def from_env(cls, ctx, timeout=120.0):
        """Communicator for the rank the launcher made this process (RANK / WORLD_SIZE), on `ctx`'s GPU."""
        rank, world = env_rank_world()
        return cls(ctx, exchange_unique_id(rank, world, timeout=timeout), rank, world)

    def rccl_origin(self):
        s = C.c_char_p()
        L.check(self.ctx.lib.r3d_comm_info(self.handle, None, None, C.byref(s)))
        return (s.value or b"").decode()

    def rccl_report(self):
        """What the communicator says about itself -- ncclCommCount / ncclCommUserRank / ncclCommCuDevice / ncclGetVersion
        (-1 where the bound library lacks the call) -- beside what the caller passed in."""
        v = [C.c_int(-1) for _ in range(4)]
        L.check(self.ctx.lib.r3d_comm_rccl_report(self.handle, *[C.byref(x) for x in v]))
        return {"world": v[0].value, "rank": v[1].value, "device": v[2].value, "version": v[3].value,
                "world_passed_in": self.world, "origin": self.rccl_origin()}

    @staticmethod
    def _counts(counts, world):
        a = np.ascontiguousarray(counts, dtype=np.int64)
        if a.shape != (world,):
            raise ValueError("need one count per rank")
        return a

    def allgather(self, d_send, byte_counts, d_recv, algo=GATHER_AUTO):
        c = self._counts(byte_counts, self.world)
        L.check(self.ctx.lib.r3d_comm_allgather(self.handle, d_send, c.ctypes.data, d_recv, int(algo)))

    def allgather_xyz(self, d_shard, points_per_rank, dtype, d_full, algo=GATHER_AUTO):
        from .device import xyz_code
        c = self._counts(points_per_rank, self.world)
        L.check(self.ctx.lib.r3d_allgather_xyz(self.handle, d_shard, c.ctypes.data, xyz_code(dtype), d_full, int(algo)))

    def allgather_inputs(self, d_depth, depth_dtype, frames_per_rank, height, width, d_pose, d_depth_all, d_pose_all,
                         algo=GATHER_AUTO):
        from .device import depth_code
        c = self._counts(frames_per_rank, self.world)
        L.check(self.ctx.lib.r3d_allgather_inputs(self.handle, d_depth, depth_code(depth_dtype), c.ctypes.data, int(height),
                                                  int(width), d_pose, d_depth_all, d_pose_all, int(algo)))

    def allreduce_sum_f64(self, d_buf, n):
        L.check(self.ctx.lib.r3d_comm_allreduce_sum_f64(self.handle, d_buf, int(n)))

    def barrier(self):
        """Every rank has reached this point (a one-number all-reduce, then a stream sync)."""
        buf = self.ctx.alloc(8).upload(np.zeros(1))
        try:
            self.allreduce_sum_f64(buf.ptr, 1)
            self.ctx.sync()
        finally:
            buf.free()

    def close(self):
        if self.handle:
            self.ctx.lib.r3d_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
