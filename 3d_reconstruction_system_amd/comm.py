"""r3d_comm: the multi-GPU exchange step through the C ABI (RCCL bound by libr3d_hip.so itself, no torch needed).

    id = Comm.unique_id()            # rank 0; ship the 128 bytes to every rank (file, pipe, MPI, torch broadcast ...)
    comm = Comm(ctx, id, rank, world)
    comm.allgather(d_send_ptr, byte_counts, d_recv_ptr)      # unequal shards, rank order, async on ctx's stream

One process per GPU (RCCL refuses two ranks on one device).
"""
import atexit
import ctypes as C
import weakref

import numpy as np

from . import _lib as L

ID_BYTES = 128
GATHER_AUTO, GATHER_NCCL, GATHER_DIRECT = 0, 1, 2


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

    def rccl_origin(self):
        s = C.c_char_p()
        L.check(self.ctx.lib.r3d_comm_info(self.handle, None, None, C.byref(s)))
        return (s.value or b"").decode()

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

    def close(self):
        if self.handle:
            self.ctx.lib.r3d_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
