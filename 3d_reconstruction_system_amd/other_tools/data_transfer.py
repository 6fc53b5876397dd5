#!/usr/bin/env python3
"""Drop-in for the reference's other_tools/data_transfer.py (f3): prepare a COLMAP dense depth render for
the fusion path -- resize to 640x480, convert to grey, save as .npy.

    get_data(img_path, write_path)         other_tools/data_transfer.py:5-16

Host-side image preparation, not a GPU kernel.  The reference does it with OpenCV, which is absent here, so the arithmetic is
restated from OpenCV's published definitions (PARITY UNPINNED -- no reference fixture can be generated without cv2):
  * resize: nearest neighbour, as the reference's comment says and its `cv2.INTER_NEAREST` argument intends
    (data_transfer.py:8-14; note that the reference passes it in the `dst` position of cv2.resize, so with a real OpenCV the
    call either fails or silently uses the bilinear default -- the intent is restated, not that accident):
    src index = min(floor(dst index * src_size / dst_size), src_size - 1)           (cv::resize, INTER_NEAREST)
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


def bgr_to_gray(bgr):
    """cv::cvtColor(BGR2GRAY) on uint8: Q14 fixed point, round to nearest."""
    b, g, r = (bgr[..., k].astype(np.uint32) for k in range(3))
    return ((b * 1868 + g * 9617 + r * 4899 + 8192) >> 14).astype(np.uint8)


def get_data(img_path, write_path):
    try:
        import cv2
    except ImportError:
        cv2 = None
    if cv2 is not None:
        img = cv2.imread(img_path)
        if img is None:
            raise FileNotFoundError("cannot read %r" % img_path)
        depth = cv2.cvtColor(cv2.resize(img, (OUT_W, OUT_H), interpolation=cv2.INTER_NEAREST), cv2.COLOR_BGR2GRAY)
    else:
        from PIL import Image
        rgb = np.array(Image.open(img_path).convert("RGB"))      # cv2.imread yields 3-channel BGR for any input
        depth = bgr_to_gray(resize_nearest(rgb[..., ::-1], OUT_W, OUT_H))
    np.save(write_path, np.array(depth))
    return depth


def main():
    depth_path = './depth/8_nprmal.png'          # the reference's literal paths (data_transfer.py:19-20)
    npy_path = './npy/8_normal.npy'
    get_data(depth_path, npy_path)


if __name__ == '__main__':
    main()
