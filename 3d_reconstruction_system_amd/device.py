"""Context / camera / device-buffer objects over the C ABI (include/r3d.h).

One `Context` = one MI355X + one HIP stream.  Device memory can come from the library
(`Context.alloc`) or from anyone else (e.g. a torch tensor's data_ptr()): the C ABI only
sees raw addresses.
"""
import ctypes as C

import numpy as np

from . import _lib as L

_DEPTH_CODES = {np.dtype(np.uint8): L.DEPTH_U8, np.dtype(np.uint16): L.DEPTH_U16,
                np.dtype(np.float32): L.DEPTH_F32}
_XYZ_CODES = {np.dtype(np.float32): L.F32, np.dtype(np.float64): L.F64}


def depth_code(dtype):
    try:
        return _DEPTH_CODES[np.dtype(dtype)]
    except KeyError:
        raise TypeError("depth raster must be uint8, uint16 or float32 (got %s)" % np.dtype(dtype))


def xyz_code(dtype):
    try:
        return _XYZ_CODES[np.dtype(dtype)]
    except KeyError:
        raise TypeError("point clouds are float32 or float64 (got %s)" % np.dtype(dtype))


_pinned_owners = {}


class _PinnedOwner:
    def __init__(self, ctx, ptr):
        self.ctx, self.ptr = ctx, ptr


def _release_pinned(key):
    owner = _pinned_owners.pop(key, None)
    if owner is not None and owner.ctx.handle:
        try:
            owner.ctx.lib.r3d_host_free(owner.ctx.handle, owner.ptr)
        except Exception:
            pass


class DeviceBuffer:
    """A library-owned HBM allocation.  `ptr` is the raw device address."""

    def __init__(self, ctx, nbytes):
        self.ctx = ctx
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        L.check(ctx.lib.r3d_dev_alloc(ctx.handle, self.nbytes, C.byref(p)))
        self.ptr = p.value

    def upload(self, host):
        host = np.ascontiguousarray(host)
        assert host.nbytes <= self.nbytes
        L.check(self.ctx.lib.r3d_memcpy_h2d(self.ctx.handle, self.ptr, host.ctypes.data, host.nbytes))
        self.ctx.sync()  # `host` may be a temporary
        return self

    def download(self, dtype, count):
        out = np.empty(count, dtype=dtype)
        assert out.nbytes <= self.nbytes
        L.check(self.ctx.lib.r3d_download(self.ctx.handle, out.ctypes.data, self.ptr, out.nbytes))   # synchronous
        return out

    def free(self):
        if self.ptr:
            L.check(self.ctx.lib.r3d_dev_free(self.ctx.handle, self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            if self.ptr and self.ctx.handle:
                self.ctx.lib.r3d_dev_free(self.ctx.handle, self.ptr)
        except Exception:
            pass


class Context:
    """r3d_ctx wrapper.  stream=None: the ctx owns a private non-blocking stream.  stream=<int>: launch on
    that raw hipStream_t, e.g. torch.cuda.current_stream().cuda_stream (0 = the device's default stream)."""

    def __init__(self, device=0, stream=None):
        self.lib = L.load()
        h = C.c_void_p()
        if stream is None:
            L.check(self.lib.r3d_ctx_create(int(device), None, 0, C.byref(h)))
        else:
            if int(stream) != 0:
                L.require_single_hip_runtime("launching on another component's HIP stream")
            L.check(self.lib.r3d_ctx_create(int(device), C.c_void_p(int(stream)), L.CTX_EXTERNAL_STREAM, C.byref(h)))
        self.handle = h.value
        self.device = int(device)
        self._cameras = {}
        self._children = []   # weak references to objects holding library handles tied to this ctx

    def adopt(self, obj):
        """Register an object with a close() method: it is closed before the context goes away."""
        import weakref
        self._children.append(weakref.ref(obj))

    def close(self):
        if self.handle:
            for ref in self._children:
                obj = ref()
                if obj is not None:
                    try:
                        obj.close()
                    except Exception:
                        pass
            self._children = []
            for cam in list(self._cameras.values()):
                cam.close()
            self._cameras.clear()
            self.lib.r3d_ctx_destroy(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        L.check(self.lib.r3d_ctx_sync(self.handle))

    def stream_handle(self):
        """The raw hipStream_t this context launches on (0 = the device's default stream)."""
        s = C.c_void_p()
        L.check(self.lib.r3d_ctx_stream(self.handle, C.byref(s)))
        return s.value or 0

    def set_tuning(self, key, value):
        L.check(self.lib.r3d_ctx_set_tuning(self.handle, key.encode(), int(value)))

    def get_tuning(self, key):
        v = C.c_int()
        L.check(self.lib.r3d_ctx_get_tuning(self.handle, key.encode(), C.byref(v)))
        return v.value

    def inputs_fresh(self):
        """Tell the library that a FOREIGN producer (torch, another library) has rewritten input buffers of this device in
        place: no raster is presumed to sit in the Infinity Cache any more, so the next fused launch stages its inputs again.
        (Writes that go through the library -- uploads, the host pipeline, r3d_comm receives -- are tracked by themselves.)"""
        self.set_tuning("fuse_inputs_fresh", 1)

    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def pinned_empty(self, shape, dtype):
        """NumPy array backed by page-locked host memory (r3d_host_alloc): the *_host entry points DMA
        straight into it instead of staging.  Freed when the array (and its views) are garbage collected."""
        dtype = np.dtype(dtype)
        count = int(np.prod(shape))
        nbytes = max(count * dtype.itemsize, 16)
        p = C.c_void_p()
        L.check(self.lib.r3d_host_alloc(self.handle, nbytes, C.byref(p)))
        buf = (C.c_char * nbytes).from_address(p.value)
        owner = _PinnedOwner(self, p.value)
        arr = np.frombuffer(buf, dtype=dtype, count=count).reshape(shape)
        _pinned_owners[id(buf)] = owner
        import weakref
        weakref.finalize(buf, _release_pinned, id(buf))
        return arr

    def timer_start(self):
        L.check(self.lib.r3d_timer_start(self.handle))

    def timer_stop(self):
        ms = C.c_float()
        L.check(self.lib.r3d_timer_stop(self.handle, C.byref(ms)))
        return ms.value

    def camera(self, height, width, fx, fy, cx, cy):
        key = (int(height), int(width), float(fx), float(fy), float(cx), float(cy))
        cam = self._cameras.get(key)
        if cam is None:
            cam = Camera(self, *key)
            self._cameras[key] = cam
        return cam


class Camera:
    """r3d_camera wrapper: intrinsics + the fp64 ray tables u[i]=(i-cx)/fx, v[j]=(j-cy)/fy in HBM."""

    def __init__(self, ctx, height, width, fx, fy, cx, cy):
        self.ctx = ctx
        self.height, self.width = int(height), int(width)
        self.fx, self.fy, self.cx, self.cy = float(fx), float(fy), float(cx), float(cy)
        h = C.c_void_p()
        L.check(ctx.lib.r3d_camera_create(ctx.handle, self.height, self.width, self.fx, self.fy, self.cx,
                                          self.cy, C.byref(h)))
        self.handle = h.value

    def close(self):
        if self.handle:
            self.ctx.lib.r3d_camera_destroy(self.handle)
            self.handle = None


_default_ctx = {}


def default_context(device=0):
    """Process-wide context per device, created on first use (raises if no MI355X is visible)."""
    ctx = _default_ctx.get(device)
    if ctx is None or ctx.handle is None:
        ctx = Context(device)
        _default_ctx[device] = ctx
    return ctx
