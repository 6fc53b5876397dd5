#!/usr/bin/env python3
"""Drop-in for the reference's other_tools/data_transfer.py (f3): prepare a COLMAP dense depth render for
the fusion path -- resize to 640x480, convert to grey, save as .npy.

    get_data(img_path, write_path)         other_tools/data_transfer.py:5-16

Host-side image preparation, not a GPU kernel.  The reference does it with OpenCV, which is absent here, so the arithmetic is
restated from OpenCV's sources (PARITY UNPINNED -- no reference fixture can be generated without cv2):
  * resize: the reference calls `cv2.resize(img, (640, 480), cv2.INTER_NEAREST)` (data_transfer.py:8) -- the constant sits in
    the `dst` position, where the Python binding turns the integer 0 into a small matrix that resize() reallocates; the
    interpolation stays at its default, INTER_LINEAR.  So the reference's files are BILINEAR resizes, whatever its comment
    says, and that is what get_data() restates by default (interpolation="as_called"):
      - 8-bit bilinear in OpenCV's fixed point (resize.cpp): source coordinate (d + 0.5) * scale - 0.5 in float, weights as
        round-to-even shorts of w * 2048, a horizontal pass into int32, then
        dst = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
      - an exact 2 x 2 reduction is switched to the box filter by cv::resize itself: (a + b + c + d + 2) >> 2.
    interpolation="nearest" gives what the comment intends: src index = min(floor(dst index * src / dst), src - 1).
  * grey: cv::cvtColor BGR2GRAY for 8-bit = (B*1868 + G*9617 + R*4899 + 8192) >> 14   (0.114 / 0.587 / 0.299 in Q14)
When cv2 is importable it is used instead, with exactly the reference's calls.
"""
import numpy as np

OUT_W, OUT_H = 640, 480


def resize_nearest(img, width, height):
    """cv::resize(..., INTER_NEAREST) index rule on an [H, W, ...] array."""
    h, w = img.shape[:2]
    xs = np.minimum(np.floor(np.arange(width) * (w / width)).astype(np.int64), w - 1)
    ys = np.minimum(np.floor(np.arange(height) * (h / height)).astype(np.int64), h - 1)
    return img[ys][:, xs]


def _linear_taps(dst_n, src_n):
    """cv::resize's INTER_LINEAR setup for one axis: (first source index, Q11 weights [dst_n, 2]); indices may be -1 or
    src_n - 1, the callers clamp the two taps the way each pass of OpenCV does."""
    scale = 1.0 / (dst_n / src_n)                                   # scale_x = 1 / inv_scale_x, both doubles
    f = ((np.arange(dst_n) + 0.5) * scale - 0.5).astype(np.float32)
    s0 = np.floor(f).astype(np.int64)
    f = (f - s0.astype(np.float32)).astype(np.float32)
    return s0, f


def _q11(w):
    # saturate_cast<short>(w * INTER_RESIZE_COEF_SCALE): float product, rounded half to even
    return np.rint(w.astype(np.float32) * np.float32(2048.0)).astype(np.int64)


def resize_linear_u8(img, width, height):
    """cv::resize(..., INTER_LINEAR) on an 8-bit [H, W(, C)] array, in OpenCV's fixed-point arithmetic (see the module
    docstring); the exact 2 x 2 reduction takes the box filter as cv::resize does."""
    img = np.asarray(img)
    if img.dtype != np.uint8:
        raise TypeError("8-bit images only")
    h, w = img.shape[:2]
    if w == 2 * width and h == 2 * height:
        a = img.astype(np.int64)
        return ((a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    sx, fx = _linear_taps(width, w)
    # horizontal taps: left of the image -> the first pixel alone, at or beyond the last pixel -> the last pixel alone
    fx = np.where(sx < 0, np.float32(0), fx)
    sx = np.where(sx < 0, 0, sx)
    fx = np.where(sx >= w - 1, np.float32(0), fx)
    sx = np.where(sx >= w - 1, w - 1, sx)
    a0, a1 = _q11(np.float32(1.0) - fx), _q11(fx)
    x1 = np.minimum(sx + 1, w - 1)                                   # weight 0 wherever this clamp acts
    src = img.astype(np.int64)
    shape = (1, width) + (1,) * (img.ndim - 2)
    rows = src[:, sx] * a0.reshape(shape) + src[:, x1] * a1.reshape(shape)          # int32 in OpenCV: at most 255 * 2048
    sy, fy = _linear_taps(height, h)
    b0, b1 = _q11(np.float32(1.0) - fy), _q11(fy)                    # vertical weights are NOT reset at the borders ...
    y0, y1 = np.clip(sy, 0, h - 1), np.clip(sy + 1, 0, h - 1)        # ... the two rows are clamped instead
    shape = (height, 1) + (1,) * (img.ndim - 2)
    out = (((b0.reshape(shape) * (rows[y0] >> 4)) >> 16) + ((b1.reshape(shape) * (rows[y1] >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def bgr_to_gray(bgr):
    """cv::cvtColor(BGR2GRAY) on uint8: Q14 fixed point, round to nearest."""
    b, g, r = (bgr[..., k].astype(np.uint32) for k in range(3))
    return ((b * 1868 + g * 9617 + r * 4899 + 8192) >> 14).astype(np.uint8)


def get_data(img_path, write_path, interpolation="as_called"):
    """interpolation: "as_called" = what the reference's call does under OpenCV (bilinear, see above), "nearest" = what its
    comment intends."""
    if interpolation not in ("as_called", "nearest"):
        raise ValueError("interpolation is 'as_called' or 'nearest'")
    try:
        import cv2
    except ImportError:
        cv2 = None
    if cv2 is not None:
        img = cv2.imread(img_path)
        if img is None:
            raise FileNotFoundError("cannot read %r" % img_path)
        if interpolation == "nearest":
            small = cv2.resize(img, (OUT_W, OUT_H), interpolation=cv2.INTER_NEAREST)
        else:
            small = cv2.resize(img, (OUT_W, OUT_H), cv2.INTER_NEAREST)          # the reference's call, argument position and all
        depth = cv2.cvtColor(small, cv2.COLOR_BGR2GRAY)
    else:
        from PIL import Image
        rgb = np.array(Image.open(img_path).convert("RGB"))      # cv2.imread yields 3-channel BGR for any input
        resize = resize_nearest if interpolation == "nearest" else resize_linear_u8
        depth = bgr_to_gray(resize(rgb[..., ::-1], OUT_W, OUT_H))
    np.save(write_path, np.array(depth))
    return depth


def main():
    depth_path = './depth/8_nprmal.png'          # the reference's literal paths (data_transfer.py:19-20)
    npy_path = './npy/8_normal.npy'
    get_data(depth_path, npy_path)


if __name__ == '__main__':
    main()
