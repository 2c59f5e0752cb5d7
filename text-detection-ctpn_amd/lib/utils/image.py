"""Host image helpers standing in for the three cv2 calls on the demo path (cv2 is not installed in this image):
cv2.resize(..., INTER_LINEAR) (reference ctpn/demo.py:25,51; lib/fast_rcnn/test.py:23), cv2.imread / imwrite
(demo.py:52,59). PARITY UNPINNED for non-identity resizes: OpenCV 3.4's fixed-point uint8 path is not
reproduced bit for bit; the float path follows the documented half-pixel-centre bilinear rule. All benchmark and
parity configs feed images already at network resolution, where both reference resizes are the identity.
"""
import numpy as np


def resize_bilinear(im, fx, fy):
    """INTER_LINEAR, dsize = round(src * f), sample position (i + 0.5) / f - 0.5, edge-clamped."""
    h, w = im.shape[:2]
    nh, nw = int(round(h * fy)), int(round(w * fx))
    if nh == h and nw == w:
        return im.copy()
    ys = (np.arange(nh) + 0.5) * (h / float(nh)) - 0.5
    xs = (np.arange(nw) + 0.5) * (w / float(nw)) - 0.5
    y0 = np.floor(ys).astype(np.int64)
    x0 = np.floor(xs).astype(np.int64)
    wy = (ys - y0).astype(np.float32)
    wx = (xs - x0).astype(np.float32)
    y0c, y1c = np.clip(y0, 0, h - 1), np.clip(y0 + 1, 0, h - 1)
    x0c, x1c = np.clip(x0, 0, w - 1), np.clip(x0 + 1, 0, w - 1)
    src = im.astype(np.float32)
    if src.ndim == 2:
        src = src[:, :, None]
    top = src[y0c][:, x0c] * (1 - wx)[None, :, None] + src[y0c][:, x1c] * wx[None, :, None]
    bot = src[y1c][:, x0c] * (1 - wx)[None, :, None] + src[y1c][:, x1c] * wx[None, :, None]
    out = top * (1 - wy)[:, None, None] + bot * wy[:, None, None]
    if im.ndim == 2:
        out = out[:, :, 0]
    if im.dtype == np.uint8:
        return np.clip(np.rint(out), 0, 255).astype(np.uint8)
    return out.astype(im.dtype)


def imread(path):
    """BGR uint8 HWC like cv2.imread; needs Pillow."""
    from PIL import Image
    with Image.open(path) as f:
        rgb = np.asarray(f.convert("RGB"))
    return np.ascontiguousarray(rgb[:, :, ::-1])


def imwrite(path, bgr):
    from PIL import Image
    Image.fromarray(np.ascontiguousarray(bgr[:, :, ::-1])).save(path)


def draw_line(img, p0, p1, color, thickness=2):
    """Bresenham-free dense line rasteriser (enough for the annotated demo output)."""
    x0, y0 = p0
    x1, y1 = p1
    n = int(max(abs(x1 - x0), abs(y1 - y0))) + 1
    xs = np.rint(np.linspace(x0, x1, n)).astype(np.int64)
    ys = np.rint(np.linspace(y0, y1, n)).astype(np.int64)
    r = max(thickness // 2, 0)
    h, w = img.shape[:2]
    for dy in range(-r, r + 1):
        for dx in range(-r, r + 1):
            yy, xx = ys + dy, xs + dx
            ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
            img[yy[ok], xx[ok]] = color
