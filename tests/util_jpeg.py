"""Shared by tests/test_jpeg.py (CPU) and tests/test_gpu_jpeg.py: JPEG test files made in memory with Pillow, and Pillow's decode of them in
cv2.imread's channel order -- the pin of the decoder (see oracle/jpeg_ref.py's header for why Pillow stands in for cv2.imread here)."""
import io

import numpy as np
from PIL import Image


def scene(h, w, seed=0, gray=False):
    """A picture with smooth regions, hard edges, saturated colours and noise: every branch of the decoder gets exercised (long zero runs and
    EOBs, ZRL, large DC differences, clamping at 0 / 255 after the colour conversion)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    a = 128 + 100 * np.sin(xx / 7.0 + seed) * np.cos(yy / 5.0)
    b = (xx * 3 + yy * 2) % 256
    c = rng.integers(0, 256, (h, w))
    img = np.stack([a, b, c], -1)
    img[h // 4: h // 2, w // 3: w // 2] = (255, 0, 0)           # saturated patches: the conversion must clamp
    img[h // 2: 3 * h // 4, w // 2: 2 * w // 3] = (0, 255, 255)
    img[: h // 8] = 0
    img[-(h // 8 + 1):] = 255
    img = np.clip(img, 0, 255).astype(np.uint8)
    return img[..., 0] if gray else img


def encode(img, quality=90, subsampling=2, **kw):
    """subsampling: 0 = 4:4:4, 1 = 4:2:2, 2 = 4:2:0 (Pillow's codes)."""
    buf = io.BytesIO()
    im = Image.fromarray(img)
    if img.ndim == 2:
        im.save(buf, "JPEG", quality=quality, **kw)
    else:
        im.save(buf, "JPEG", quality=quality, subsampling=subsampling, **kw)
    return buf.getvalue()


def with_luma_sampling(data, hv):
    """The file with the luma sampling byte of its frame header replaced (0x12 = 4:4:0, 0x41 = 4:1:1 ...): layouts Pillow cannot write, for
    the tests of what the decoder refuses (the entropy data no longer matches: header checks only)."""
    i = max(data.find(b"\xff\xc0"), data.find(b"\xff\xc2"))
    assert i > 0 and data[i + 9] == 3
    return data[:i + 11] + bytes([hv]) + data[i + 12:]


def pillow_bgr(data):
    """What lib/utils/image.py's imread returns for the file: Pillow's decode, BGR (gray files replicated, like cv2.IMREAD_COLOR)."""
    rgb = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
    return np.ascontiguousarray(rgb[..., ::-1])


# (h, w, quality, subsampling, gray, extra save arguments): whole-MCU sizes, odd sizes in both directions (partial MCUs, the replicated chroma
# edge), one MCU only, one pixel, low quality (16-bit-free but long runs), quality 100 (all-ones tables, large coefficients), optimised
# Huffman tables (code lengths the standard tables do not have), restart intervals, gray, progressive
CASES = [
    (48, 64, 90, 2, False, {}),
    (37, 53, 90, 2, False, {}),
    (40, 40, 75, 0, False, {}),
    (33, 47, 95, 0, False, {}),
    (16, 16, 50, 2, False, {}),
    (1, 1, 90, 2, False, {}),
    (20, 3, 95, 2, False, {}),                   # downsampled width <= 2: libjpeg replicates the chroma instead of filtering it
    (21, 4, 95, 2, False, {}),
    (3, 2, 95, 2, False, {}),
    (4, 5, 95, 2, False, {}),                    # ... and 3 is the narrowest filtered one
    (9, 17, 100, 2, False, {}),
    (50, 70, 30, 2, False, {"optimize": True}),
    (45, 61, 85, 2, False, {"restart_marker_blocks": 3}),
    (45, 61, 85, 0, False, {"restart_marker_rows": 1}),
    (31, 42, 90, 2, True, {}),
    (24, 24, 60, 2, True, {"restart_marker_blocks": 2}),
    # progressive files (SOF2): libjpeg's default scan script -- an interleaved DC scan, AC bands per component, then refinement scans for both
    (48, 64, 90, 2, False, {"progressive": True}),
    (37, 53, 75, 0, False, {"progressive": True, "optimize": True}),
    (45, 61, 85, 2, False, {"progressive": True, "restart_marker_blocks": 3}),
    (31, 42, 90, 2, True, {"progressive": True}),
    (20, 3, 95, 2, False, {"progressive": True}),
    # 4:2:2 (chroma halved horizontally only: h2v1 upsampling), whole and partial MCUs, the narrow images whose chroma is replicated
    (48, 64, 90, 1, False, {}),
    (37, 53, 85, 1, False, {"optimize": True}),
    (21, 4, 95, 1, False, {}),
    (19, 5, 95, 1, False, {}),
    (45, 61, 80, 1, False, {"progressive": True, "restart_marker_blocks": 2}),
]


def case_id(c):
    h, w, q, s, g, kw = c
    return "%dx%d-q%d-%s%s" % (h, w, q, "gray" if g else ("444", "422", "420")[s], "".join("-" + k for k in kw))
