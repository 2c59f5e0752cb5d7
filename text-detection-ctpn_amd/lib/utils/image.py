"""Image helpers standing in for the cv2 calls on the demo path (cv2 is not installed in this image):
cv2.resize(..., INTER_LINEAR) (reference ctpn/demo.py:25,51; lib/fast_rcnn/test.py:23) runs on the GPU through
ctpn_resize (csrc/preprocess.hip; OpenCV 3.4's algorithm restated, uint8 fixed-point and float32 paths -- parity with
the real cv2 UNPINNED, bit-exact against oracle/resize_ref.py); cv2.imread / imwrite (demo.py:52,59) go through Pillow.
There is no host implementation of the resize here: without a GPU a non-identity resize raises.
"""
import numpy as np

from ..._binding import resize_dims, resize_linear


def resize_bilinear(im, fx, fy, device_id=0):
    """cv2.resize(im, None, None, fx=fx, fy=fy, interpolation=cv2.INTER_LINEAR); (h,w,3) uint8 or float32."""
    im = np.asarray(im)
    h, w = im.shape[:2]
    if resize_dims(h, w, fx, fy) == (h, w) and fx == 1.0 and fy == 1.0:
        return im.copy()                       # every sample falls on a source pixel with weight 1: exact copy, like cv2
    return resize_linear(im, fx, fy, device_id)


def _oriented(f):
    """cv2.imread turns a JPEG by its EXIF orientation tag (OpenCV >= 3.1 unless IMREAD_IGNORE_ORIENTATION; the reference calls it with the
    default flags, ctpn/demo.py:59); Pillow leaves that to the caller. JPEG only: that is where OpenCV 3.x looks for EXIF."""
    if f.format == "JPEG" and f.getexif().get(0x0112, 1) != 1:
        from PIL import ImageOps
        return ImageOps.exif_transpose(f)
    return f


def open_rgb(path):
    """(h, w, 3) RGB uint8 of an image file the way cv2.imread(path) sees it (orientation applied, alpha dropped), before the BGR flip."""
    from PIL import Image
    with Image.open(path) as f:
        if f.format == "PNG" and f.mode in ("I", "I;16", "I;16B"):
            # a 16-bit gray PNG: libpng hands cv2.imread(IMREAD_COLOR) the high byte of every sample (png_set_strip_16); Pillow's
            # convert("RGB") would clip at 255 instead
            g = (np.asarray(f).astype(np.uint16) >> 8).astype(np.uint8)
            return np.repeat(g[:, :, None], 3, axis=2)
        return np.asarray(_oriented(f).convert("RGB"))


def image_size(path):
    """(h, w) of what imread(path) returns, from the file header only (no pixel decode)."""
    from PIL import Image
    with Image.open(path) as f:
        w, h = f.size
        if f.format == "JPEG" and f.getexif().get(0x0112, 1) in (5, 6, 7, 8):
            w, h = h, w
    return h, w


def imread(path):
    """BGR uint8 HWC like cv2.imread(path). PNG files go through the library's decoder (ctpn_png_decode: host code by the nature of the format,
    libdeflate + row filters, byte-equal to Pillow's result at half its time); what it does not take (16-bit PNG) and every other format
    through Pillow."""
    if str(path).lower().endswith(".png"):
        from ..._binding import CtpnError, png_decode
        try:
            with open(path, "rb") as f:
                return png_decode(f.read())
        except CtpnError:
            pass
    return np.ascontiguousarray(open_rgb(path)[:, :, ::-1])


def imwrite(path, bgr):
    """cv2.imwrite(path, img) with its default parameters (reference ctpn/demo.py:52): JPEG at quality 95 (IMWRITE_JPEG_QUALITY's default; Pillow's
    own default would be 75) with libjpeg's default 4:2:0 sampling, PNG at compression level 1 (3.x: IMWRITE_PNG_COMPRESSION = 1 -- lossless
    either way). The format follows the file name's extension, as in OpenCV."""
    from PIL import Image
    im = Image.fromarray(np.ascontiguousarray(bgr[:, :, ::-1]))
    ext = path.rsplit(".", 1)[-1].lower() if "." in path else ""
    if ext in ("jpg", "jpeg", "jpe"):
        im.save(path, quality=95, subsampling=2)
    elif ext == "png":
        im.save(path, compress_level=1)
    else:
        im.save(path)


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
