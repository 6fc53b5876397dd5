"""Files on either side of the hot path, in the reference's formats.

  depth PNG  -> uint8 raster      cv.imread(..., IMREAD_GRAYSCALE)        c2w:160
  camera txt `X,Y,Z\\n`           gentxtcord                               c2w:73-83, p2c:33-43
  world txt                       get_pointdata / local_world             c2w:103-104, icp:87-97
  ASCII PLY                       genply                                  c2w:112-134, icp:46-68

Writers go through the library's native multi-threaded formatter (byte-identical to Python's
"%.4f" and repr()); readers are NumPy.
"""
import ctypes as C
import os

import numpy as np

from . import _lib as L
from .device import xyz_code


GRAY_RULES = {"opencv_png": 0, "cvtcolor": 1}


class UnsupportedDepthFormat(L.R3DError):
    """A depth file this build cannot turn into OpenCV's IMREAD_GRAYSCALE raster bit for bit (R3D_ERR_UNSUPPORTED)."""

    def __init__(self, message):
        L.R3DError.__init__(self, L.ERR_UNSUPPORTED, message)


def _is_png(path):
    try:
        with open(path, "rb") as f:
            return f.read(8) == b"\x89PNG\r\n\x1a\n"
    except FileNotFoundError:
        raise FileNotFoundError("cannot read depth image %r" % path)


def _is_jpeg(path):
    with open(path, "rb") as f:
        return f.read(3) == b"\xff\xd8\xff"


def _kind(path):
    """'png' / 'jpeg' / 'other' by the file's first bytes ('missing' when it cannot be opened)."""
    try:
        with open(path, "rb") as f:
            head = f.read(8)
    except OSError:
        return "missing"
    if head == b"\x89PNG\r\n\x1a\n":
        return "png"
    return "jpeg" if head[:3] == b"\xff\xd8\xff" else "other"


def rgb_to_gray(rgb, rule="cvtcolor"):
    """[...,3|4] uint8 R,G,B(,A) -> [...] uint8 grey by one of OpenCV's two integer rules (include/r3d.h, R3D_GRAY_*):
    "opencv_png" = what cv.imread(<png>, IMREAD_GRAYSCALE) does (libpng's rgb_to_gray), "cvtcolor" = cv.cvtColor(BGR2GRAY)."""
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    out = np.empty(rgb.shape[:-1], np.uint8)
    L.check(L.load().r3d_rgb_to_gray_u8(rgb.ctypes.data, out.size, rgb.shape[-1], GRAY_RULES[rule], out.ctypes.data))
    return out


def _default_rule():
    return os.environ.get("R3D_GRAY_RULE", "opencv_png")


def read_depth_gray(path, rule=None, allow_pil_jpeg=None):
    """uint8 raster with the meaning of `cv.imread(path, cv.IMREAD_GRAYSCALE)` (c2w:160), without OpenCV.
      * PNG (the reference's input): decoded natively -- 8-bit grey as stored, 16-bit -> high byte, alpha dropped, colour
        converted by `rule` ("opencv_png", the default: libpng's rgb_to_gray as OpenCV's PNG reader requests it; or
        "cvtcolor"; env R3D_GRAY_RULE).  PIL's 'L' weights are neither and are not used.
      * JPEG (AirSim stores depth as 3-channel JPG): OpenCV asks libjpeg itself for grey output -- the luma component through
        libjpeg's integer IDCT.  Sequential Huffman files (what cv.imwrite and PIL write) are decoded natively by a restatement
        of exactly that (csrc/r3d_jpeg.cpp; byte-identical to libjpeg-turbo, tests/test_host_logic.py).  Progressive / CMYK /
        12-bit files are refused unless allow_pil_jpeg=True / R3D_ALLOW_PIL_JPEG=1 lets PIL decode them.
      * other formats PIL reads (BMP, TIFF, ...): decoded to RGB by PIL (lossless formats: the same samples), converted by
        the cvtColor rule, which is what imread does for them."""
    path = os.fspath(path)
    try:
        import cv2                                   # the real thing, where it exists
        img = cv2.imread(path, cv2.IMREAD_GRAYSCALE)
        if img is None:
            raise FileNotFoundError("cannot read depth image %r" % path)
        return np.ascontiguousarray(img)
    except ImportError:
        pass
    rule = rule or _default_rule()
    if _is_png(path):
        lib = L.load()
        h, w = C.c_int(), C.c_int()
        rc = lib.r3d_png_gray8_info(os.fsencode(path), C.byref(h), C.byref(w))
        if rc == L.OK:
            img = np.empty((h.value, w.value), np.uint8)
            arr = (C.c_char_p * 1)(os.fsencode(path))
            rc = lib.r3d_png_gray8_decode_batch(arr, 1, img.ctypes.data, h.value, w.value, GRAY_RULES[rule])
            if rc == L.ERR_UNSUPPORTED:                 # e.g. real colour in a gamma-tagged file: the message says what to do
                raise UnsupportedDepthFormat(L.last_error())
            L.check(rc)
            return img
        if rc != L.ERR_UNSUPPORTED:
            L.check(rc)
        raise UnsupportedDepthFormat("%r: %s.  Palette / interlaced / sub-byte PNGs are not decoded natively and PIL's grey "
                                     "conversion is not OpenCV's; re-save the file as a plain 8/16-bit PNG" % (path, L.last_error()))
    native_said = None
    if _is_jpeg(path):
        lib = L.load()
        h, w = C.c_int(), C.c_int()
        rc = lib.r3d_jpeg_gray_info(os.fsencode(path), C.byref(h), C.byref(w))
        if rc == L.OK:
            img = np.empty((h.value, w.value), np.uint8)
            arr = (C.c_char_p * 1)(os.fsencode(path))
            L.check(lib.r3d_jpeg_gray_decode_batch(arr, 1, img.ctypes.data, h.value, w.value))
            return img
        if rc != L.ERR_UNSUPPORTED:
            L.check(rc)
        native_said = L.last_error()
    from PIL import Image
    pil = Image.open(path)
    if pil.format in ("JPEG", "MPO"):
        if allow_pil_jpeg is None:
            allow_pil_jpeg = os.environ.get("R3D_ALLOW_PIL_JPEG", "0") not in ("", "0")
        if not allow_pil_jpeg:
            raise UnsupportedDepthFormat(
                "%s.  The native decoder takes sequential Huffman JPEGs (what OpenCV and PIL write) and restates libjpeg's grey output for "
                "them; for this flavour it cannot promise OpenCV's IMREAD_GRAYSCALE bytes.  Fallback: read_depth_gray(path, "
                "allow_pil_jpeg=True) or R3D_ALLOW_PIL_JPEG=1 decodes it through PIL (libjpeg asked for greyscale output, the request "
                "OpenCV makes); or convert the depth maps to PNG" % (native_said or repr(path)))
        pil.draft("L", pil.size)
        return np.ascontiguousarray(np.array(pil.convert("L") if pil.mode != "L" else pil))
    if pil.mode == "L":
        return np.ascontiguousarray(np.array(pil))
    if pil.mode.startswith("I;16") or pil.mode == "I":
        return np.ascontiguousarray((np.array(pil).astype(np.uint32) >> 8).clip(0, 255).astype(np.uint8))
    return rgb_to_gray(np.array(pil.convert("RGB")), "cvtcolor")


def _jpeg_backend(backend):
    """'native' (csrc/r3d_jpeg.cpp on the library's host threads) unless the caller asks for 'pil' (libjpeg-turbo through PIL on
    a Python thread pool: the same bytes, tests/test_host_logic.py).  Measured round 5: per thread libjpeg-turbo is
    ~2x faster, but PIL's grey (draft) decode does not scale over threads at all and its colour decode scales to the native
    pool's rate at best (8 threads, 1080p: grey 9.1 vs 2.2 ms/file, colour 3.7 + copy vs 3.9), so native stays the default."""
    backend = backend or "native"
    if backend not in ("pil", "native"):
        raise ValueError("JPEG backend must be 'native' or 'pil'")
    return backend


def _pil_jpeg_batch(paths, out, colour):
    """A batch of baseline JPEGs through PIL's libjpeg-turbo on a thread pool (the decoder releases the GIL): grey = libjpeg
    asked for greyscale output (draft 'L': the luma plane through the integer IDCT, OpenCV's IMREAD_GRAYSCALE request), colour
    = Image.open(...).convert('RGB') (p2c:58-60).  Returns False when a file is not what the batch expects (size, flavour):
    the caller then takes the native / file-by-file way, which also owns the error messages."""
    from concurrent.futures import ThreadPoolExecutor
    from PIL import Image
    want = out.shape[1:3]

    def one(k):
        with Image.open(paths[k]) as im:
            if im.format != "JPEG" or im.size != (want[1], want[0]) or im.mode not in ("L", "RGB") or \
                    (not colour and im.info.get("progressive")):     # grey from a progressive file: read_depth_gray's rule decides
                return False
            if not colour:
                im.draft("L", im.size)
            im = im.convert("RGB" if colour else "L")
            if im.size != (want[1], want[0]):
                return False
            out[k] = np.asarray(im)
        return True
    workers = max(1, min(len(paths), len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count() or 1, 64))
    if workers == 1:
        return all(one(k) for k in range(len(paths)))
    with ThreadPoolExecutor(workers) as pool:
        return all(pool.map(one, range(len(paths))))


def read_depth_batch(paths, out=None, rule=None, jpeg_backend=None):
    """[F,H,W] raster batch for a list of depth files with IMREAD_GRAYSCALE meaning (c2w:160).  PNGs of any supported
    flavour -- 8/16-bit, grey or colour (read_depth_gray has the rules) -- are decoded by the library's host threads straight
    into one contiguous (optionally pinned) buffer; sequential JPEGs by the library's own decoder
    (jpeg_backend='pil': libjpeg-turbo through PIL on a thread pool, same bytes); other formats go file by file."""
    paths = [os.fspath(p) for p in paths]
    if not paths:
        return np.empty((0, 0, 0), np.uint8)
    rule = rule or _default_rule()
    lib = L.load()
    h, w = C.c_int(), C.c_int()
    if not os.path.exists(paths[0]):
        raise FileNotFoundError("cannot read depth image %r" % paths[0])
    kinds = {_kind(p) for p in paths}
    jpeg = kinds == {"jpeg"}
    use_pil = jpeg and _jpeg_backend(jpeg_backend) == "pil"
    if jpeg:
        rc = lib.r3d_jpeg_gray_info(os.fsencode(paths[0]), C.byref(h), C.byref(w))
    elif kinds == {"png"}:
        rc = lib.r3d_png_gray8_info(os.fsencode(paths[0]), C.byref(h), C.byref(w))
    else:
        rc = L.ERR_UNSUPPORTED                                     # other or mixed formats (or a missing file): file by file below
    if rc == L.OK:
        shape = (len(paths), h.value, w.value)
        if out is None:
            out = np.empty(shape, np.uint8)
        elif out.shape != shape or out.dtype != np.uint8 or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous uint8 array of shape %s" % (shape,))
        if use_pil and _pil_jpeg_batch(paths, out, colour=False):
            return out
        arr = (C.c_char_p * len(paths))(*[os.fsencode(p) for p in paths])
        if jpeg:
            rc = lib.r3d_jpeg_gray_decode_batch(arr, len(paths), out.ctypes.data, h.value, w.value)
        else:
            rc = lib.r3d_png_gray8_decode_batch(arr, len(paths), out.ctypes.data, h.value, w.value, GRAY_RULES[rule])
        if rc == L.OK:
            return out
        if rc != L.ERR_UNSUPPORTED:
            for p in paths:
                if not os.path.exists(p):
                    raise FileNotFoundError("cannot read depth image %r" % p)
            L.check(rc)
    elif rc != L.ERR_UNSUPPORTED:
        L.check(rc)
    rasters = [read_depth_gray(p, rule) for p in paths]          # a flavour the batch decoder does not take: file by file
    for p, r in zip(paths, rasters):
        if r.shape != rasters[0].shape:
            raise ValueError("depth %s is %s, expected %s: all frames of one pose file share a camera"
                             % (p, r.shape, rasters[0].shape))
    return np.stack(rasters)


def read_rgb_batch(paths, out=None, jpeg_backend=None):
    """[F,H,W,3] uint8 (R,G,B) for a list of colour images -- the colour planes of the RGBD path (fuse_frames_rgb).
    8-bit RGB / RGBA / grey PNGs are decoded by the library's host threads into one contiguous (optionally pinned) buffer;
    sequential YCbCr / grey JPEGs (4:4:4, 4:2:2, 4:2:0) give the bytes PIL's Image.open gives, which is what the reference
    reads colour with (p2c:58-60) -- through the library's decoder, or PIL itself on a thread pool (jpeg_backend='pil', same
    bytes); anything else (BMP, TIFF, mixed lists, ...) goes through PIL file by file."""
    paths = [os.fspath(p) for p in paths]
    if not paths:
        return np.empty((0, 0, 0, 3), np.uint8)
    lib = L.load()
    h, w, ch = C.c_int(), C.c_int(), C.c_int()
    kinds = {_kind(p) for p in paths}
    for p in paths:
        if not os.path.exists(p):
            raise FileNotFoundError("cannot read image %r" % p)
    jpeg = kinds == {"jpeg"}
    if jpeg:
        rc = lib.r3d_jpeg_rgb_info(os.fsencode(paths[0]), C.byref(h), C.byref(w), C.byref(ch))
    elif kinds == {"png"}:
        rc = lib.r3d_png_rgb_info(os.fsencode(paths[0]), C.byref(h), C.byref(w), C.byref(ch))
    else:
        rc = L.ERR_UNSUPPORTED                                     # other or mixed formats: PIL below, as the reference does
    use_pil = jpeg and _jpeg_backend(jpeg_backend) == "pil"
    if rc == L.OK:
        shape = (len(paths), h.value, w.value, 3)
        if out is None:
            out = np.empty(shape, np.uint8)
        elif out.shape != shape or out.dtype != np.uint8 or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous uint8 array of shape %s" % (shape,))
        if use_pil and _pil_jpeg_batch(paths, out, colour=True):
            return out
        arr = (C.c_char_p * len(paths))(*[os.fsencode(p) for p in paths])
        decode = lib.r3d_jpeg_rgb_decode_batch if jpeg else lib.r3d_png_rgb_decode_batch
        rc = decode(arr, len(paths), out.ctypes.data, h.value, w.value)
        if rc == L.OK:
            return out
        if rc != L.ERR_UNSUPPORTED:
            L.check(rc)
    elif rc != L.ERR_UNSUPPORTED:
        if not os.path.exists(paths[0]):
            raise FileNotFoundError("cannot read image %r" % paths[0])
        L.check(rc)
    from PIL import Image
    imgs = [np.array(Image.open(p).convert("RGB")) for p in paths]
    for p, r in zip(paths, imgs):
        if r.shape != imgs[0].shape:
            raise ValueError("image %s is %s, expected %s" % (p, r.shape, imgs[0].shape))
    return np.stack(imgs)


def read_depth_unchanged(path):
    """IMREAD_UNCHANGED: channels in BGR order like OpenCV (p2c:133 then takes channel 1)."""
    try:
        import cv2
        img = cv2.imread(path, cv2.IMREAD_UNCHANGED)
    except ImportError:
        from PIL import Image
        try:
            img = np.array(Image.open(path))
        except FileNotFoundError:
            img = None
        if img is not None and img.ndim == 3 and img.shape[2] >= 3:
            img = img[:, :, [2, 1, 0] + list(range(3, img.shape[2]))]
    if img is None:
        raise FileNotFoundError("cannot read image %r" % path)
    return np.ascontiguousarray(img)


def _cloud(xyz):
    xyz = np.ascontiguousarray(xyz)
    if xyz.ndim != 2 or xyz.shape[1] != 3:
        raise ValueError("cloud must be [N,3]")
    xyz_code(xyz.dtype)
    return xyz


def write_xyz_txt(path, xyz, z_raw=None, append=False):
    """`X,Y,Z\\n` per point with repr() floats.  z_raw: optional integer raster (uint8/uint16,
    N values) printed as the third column instead, like the reference's camera txt, where Z is
    `str(np.uint8)`."""
    xyz = _cloud(xyz)
    keep, zp, zc = _z_raw_args(xyz, z_raw)
    L.check(L.load().r3d_write_xyz_txt(os.fsencode(path), xyz.ctypes.data, xyz_code(xyz.dtype), xyz.shape[0],
                                       zp, zc, 1 if append else 0))


def write_xyz_txt_batch(paths, xyz, z_raw=None):
    """One `X,Y,Z\\n` file per frame: file k takes rows [k*P, (k+1)*P) of xyz ([F*P,3]; P = rows / len(paths)) -- the
    ./point/<stem>.txt files of the frame loop (c2w:163-165), written by the library's host threads, one file per thread at a
    time.  Same bytes as write_xyz_txt per file."""
    paths = [os.fspath(p) for p in paths]
    xyz = _cloud(xyz)
    if not paths:
        if xyz.shape[0]:
            raise ValueError("points but no files to put them in")
        return
    if xyz.shape[0] % len(paths):
        raise ValueError("%d points do not divide into %d files" % (xyz.shape[0], len(paths)))
    keep, zp, zc = _z_raw_args(xyz, z_raw)
    arr = (C.c_char_p * len(paths))(*[os.fsencode(p) for p in paths])
    L.check(L.load().r3d_write_xyz_txt_batch(arr, len(paths), xyz.ctypes.data, xyz_code(xyz.dtype), xyz.shape[0] // len(paths), zp, zc))


def _z_raw_args(xyz, z_raw):
    if z_raw is None:
        return None, None, 0
    z_raw = np.ascontiguousarray(z_raw).reshape(-1)
    if z_raw.shape[0] != xyz.shape[0] or z_raw.dtype not in (np.uint8, np.uint16):
        raise ValueError("z_raw must be uint8/uint16 with one value per point")
    return z_raw, z_raw.ctypes.data, (L.DEPTH_U8 if z_raw.dtype == np.uint8 else L.DEPTH_U16)


def format_xyz_txt(xyz, z_raw=None):
    """The `X,Y,Z\\n` text as bytes (same formatting as write_xyz_txt)."""
    xyz = _cloud(xyz)
    keep, zp, zc = _z_raw_args(xyz, z_raw)
    lib = L.load()
    n = C.c_size_t()
    L.check(lib.r3d_format_xyz_txt(xyz.ctypes.data, xyz_code(xyz.dtype), xyz.shape[0], zp, zc, None, 0, C.byref(n)))
    buf = C.create_string_buffer(max(n.value, 1))
    L.check(lib.r3d_format_xyz_txt(xyz.ctypes.data, xyz_code(xyz.dtype), xyz.shape[0], zp, zc, buf, n.value,
                                   C.byref(n)))
    return buf.raw[:n.value]


def format_ply(xyz):
    """The reference PLY bytes for an [N,3] cloud."""
    xyz = _cloud(xyz)
    lib = L.load()
    n = C.c_size_t()
    L.check(lib.r3d_format_ply(xyz.ctypes.data, xyz_code(xyz.dtype), xyz.shape[0], None, 0, C.byref(n)))
    buf = C.create_string_buffer(n.value)
    L.check(lib.r3d_format_ply(xyz.ctypes.data, xyz_code(xyz.dtype), xyz.shape[0], buf, n.value, C.byref(n)))
    return buf.raw[:n.value]


def write_ply(path, xyz):
    xyz = _cloud(xyz)
    L.check(L.load().r3d_write_ply(os.fsencode(path), xyz.ctypes.data, xyz_code(xyz.dtype), xyz.shape[0]))


def write_ply_binary(path, xyz):
    """f1's optional binary flag: the vertices as a standard binary_little_endian PLY (float32 x, y, z; 12 B/vertex).  NOT the
    reference's file -- an opt-in for users whose next tool reads PLY (the drop-in takes it under R3D_PLY_BINARY=1)."""
    xyz = _cloud(xyz)
    L.check(L.load().r3d_write_ply_binary(os.fsencode(path), xyz.ctypes.data, xyz_code(xyz.dtype), xyz.shape[0]))


def write_ply_rgb(path, xyz, rgb):
    """Coloured PLY in the reference layout (p2c:55-91).  rgb: [N,3] uint8 (R,G,B), or the [N] uint32 rgba words of
    fuse_frames_rgb (bytes R,G,B,0); the alpha column is the literal 0 either way."""
    xyz = np.ascontiguousarray(xyz)
    rgb = np.asarray(rgb)
    lib = L.load()
    if rgb.dtype == np.uint32 and rgb.ndim == 1:
        rgb = np.ascontiguousarray(rgb)
        if rgb.shape[0] != xyz.shape[0]:
            raise ValueError("%d colours, cloud has %d points" % (rgb.shape[0], xyz.shape[0]))
        L.check(lib.r3d_write_ply_rgba(os.fsencode(path), xyz.ctypes.data, xyz_code(xyz.dtype), rgb.ctypes.data,
                                       xyz.shape[0]))
        return
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8).reshape(-1, 3)
    if rgb.shape[0] != xyz.shape[0]:
        raise ValueError("colour image has %d pixels, cloud has %d points" % (rgb.shape[0], xyz.shape[0]))
    L.check(lib.r3d_write_ply_rgb(os.fsencode(path), xyz.ctypes.data, xyz_code(xyz.dtype), rgb.ctypes.data,
                                  xyz.shape[0]))


def _parse_rows_native(buf, offset, separator):
    """[N,3] float64 from the text in `buf[offset:]` by the library's threaded parser, or None when a line is not plain
    'x<sep>y<sep>z[...]' in the subset std::from_chars shares with float() -- the caller then parses it Python's way."""
    lib = L.load()
    n, bad = C.c_int64(), C.c_int64()
    if len(buf) - offset <= 0:
        return np.empty((0, 3))
    raw = (C.c_char * len(buf)).from_buffer(buf)          # buf: bytearray (no copy)
    ptr = C.addressof(raw) + offset
    nbytes = len(buf) - offset
    if lib.r3d_parse_xyz_text(ptr, nbytes, ord(separator), None, 0, C.byref(n), None) != L.OK:
        return None
    out = np.empty((n.value, 3), dtype=np.float64)
    rc = lib.r3d_parse_xyz_text(ptr, nbytes, ord(separator), out.ctypes.data, n.value, C.byref(n), C.byref(bad))
    return out if rc == L.OK else None


def read_xyz_txt(path):
    """[N,3] float64 from `X,Y,Z\n` lines: the first three comma-separated fields of every non-empty line, like the
    reference's data_p[0:3] (c2w:97-98, icp:76-80) -- extra fields (e.g. x,y,z,r,g,b) are ignored, a missing final
    newline is fine, a malformed line raises ValueError naming it.  Parsed by the library's host threads
    (r3d_parse_xyz_text, correctly rounded like float()); a file it declines is parsed line by line here."""
    with open(path, 'rb') as f:
        data = bytearray(f.read())
    got = _parse_rows_native(data, 0, ',')
    if got is not None:
        return got
    lines = data.decode('utf-8', errors='replace').split('\n')
    if lines and lines[-1] == '':
        lines.pop()
    out = np.empty((len(lines), 3))
    n = 0
    for no, line in enumerate(lines, 1):
        if not line.strip():
            continue
        fields = line.split(',')
        try:
            if len(fields) < 3:
                raise ValueError
            out[n] = [float(v) for v in fields[:3]]
        except ValueError:
            raise ValueError("%s line %d: expected 'x,y,z[,...]', got %r" % (path, no, line[:80])) from None
        n += 1
    return out[:n]


def read_ply(path):
    """Vertices of an ASCII PLY in the reference layout -> [N,3] float64 (the first three blank-separated numbers of each
    of the `element vertex` rows; colour columns are ignored); also reads write_ply_binary's files."""
    with open(path, 'rb') as f:
        data = bytearray(f.read())
    n, start, pos = None, None, 0
    while pos < len(data):
        e = data.find(b'\n', pos)
        e = len(data) if e < 0 else e
        s = bytes(data[pos:e]).strip()
        if s.startswith(b'element vertex'):
            n = int(s.split()[-1])
        pos = e + 1
        if s == b'end_header':
            start = pos
            break
    if n is None or start is None:
        raise ValueError("%s: not a PLY with an 'element vertex' header" % path)
    if b'format binary_little_endian' in bytes(data[:start]):       # write_ply_binary's layout: float32 x, y, z per vertex
        return np.frombuffer(bytes(data[start:start + n * 12]), dtype='<f4').reshape(-1, 3).astype(np.float64)
    got = _parse_rows_native(data, min(start, len(data)), ' ') if start < len(data) else np.empty((0, 3))
    if got is not None and got.shape[0] == n:
        return got                                   # the whole body is the vertex list (the reference's layout)
    lines = data[start:].decode('utf-8', errors='replace').split('\n')
    rows = [s.split()[:3] for s in lines[:n]]
    return np.array(rows, dtype=np.float64).reshape(-1, 3)
