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


# ---------------------------------------------------------------------------------------------------------------------------------------
# A small baseline encoder of our own, for the layouts Pillow cannot write (4:4:0 = luma 1 x 2; anything else for the refusal tests) and for
# files with an EXIF orientation: float DCT, one flat-ish quantisation table pair, and the simplest legal Huffman tables (every DC category
# a 4-bit code, every AC symbol an 8-bit code -- incomplete codes are legal JPEG). What it writes is only ever an INPUT: Pillow's decode of
# the same bytes is the pin, as for every other file here.
# ---------------------------------------------------------------------------------------------------------------------------------------
_ZZ = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56,
       57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]
_AC_SYMS = [0x00, 0xF0] + [(r << 4) | s for r in range(16) for s in range(1, 11)]


def exif_app1(orientation, big_endian=False):
    """An APP1 segment holding one IFD0 entry: Orientation (0x0112, SHORT)."""
    import struct
    e = ">" if big_endian else "<"
    tiff = (b"MM" if big_endian else b"II") + struct.pack(e + "HI", 42, 8) + struct.pack(e + "H", 1) + \
        struct.pack(e + "HHI", 0x0112, 3, 1) + struct.pack(e + "H", orientation) + b"\0\0" + struct.pack(e + "I", 0)
    body = b"Exif\0\0" + tiff
    return b"\xff\xe1" + struct.pack(">H", len(body) + 2) + body


def with_exif_orientation(data, orientation, big_endian=False):
    """The file with an EXIF APP1 segment (orientation tag only) inserted behind SOI / the JFIF segment."""
    i = 2
    if data[2:4] == b"\xff\xe0":
        i = 4 + ((data[4] << 8) | data[5])
    return data[:i] + exif_app1(orientation, big_endian) + data[i:]


def encode_custom(img, hs=1, vs=2, q=8, restart=0, orientation=None):
    """(h, w, 3) RGB uint8 (or (h, w) gray) -> baseline JPEG bytes with luma sampling hs x vs and 1 x 1 chroma. q: the quantisation step
    of the low frequencies (it grows with the frequency)."""
    import struct
    from scipy.fft import dctn
    img = np.asarray(img)
    gray = img.ndim == 2
    h, w = img.shape[:2]
    if gray:
        planes, samp = [img.astype(np.float64)], [(1, 1)]
        hs = vs = 1
    else:
        r, g, b = (img[..., k].astype(np.float64) for k in range(3))
        y = 0.299 * r + 0.587 * g + 0.114 * b
        cb = -0.168736 * r - 0.331264 * g + 0.5 * b + 128
        cr = 0.5 * r - 0.418688 * g - 0.081312 * b + 128
        ph, pw = -(-h // vs) * vs, -(-w // hs) * hs

        def down(c):
            c = np.pad(c, ((0, ph - h), (0, pw - w)), mode="edge")
            return c.reshape(ph // vs, vs, pw // hs, hs).mean(axis=(1, 3))
        planes, samp = [y, down(cb), down(cr)], [(hs, vs), (1, 1), (1, 1)]
    mcux, mcuy = -(-w // (8 * hs)), -(-h // (8 * vs))
    qt = np.array([[min(255, q + (q * (u + v)) // 2) for u in range(8)] for v in range(8)], np.int64)
    blocks = []
    for p, (ch, cv) in zip(planes, samp):
        H, W = mcuy * cv * 8, mcux * ch * 8
        p = np.pad(p, ((0, H - p.shape[0]), (0, W - p.shape[1])), mode="edge") - 128.0
        bl = p.reshape(H // 8, 8, W // 8, 8).transpose(0, 2, 1, 3)
        co = np.rint(dctn(bl, type=2, norm="ortho", axes=(2, 3)) / qt).astype(np.int64)
        co[..., 1:8, :] = np.clip(co[..., 1:8, :], -1023, 1023)
        co[..., 0, 1:] = np.clip(co[..., 0, 1:], -1023, 1023)
        blocks.append(co.reshape(H // 8, W // 8, 64))
    bits = []

    def put(code, n):
        bits.append((code, n))

    def mag(v):
        t = int(abs(int(v))).bit_length()
        return t, (int(v) if v >= 0 else int(v) + (1 << t) - 1)
    ac_code = {s: i for i, s in enumerate(_AC_SYMS)}
    out = bytearray()

    def flush_bits():
        acc = n = 0
        for code, k in bits:
            acc = (acc << k) | (code & ((1 << k) - 1))
            n += k
            while n >= 8:
                byte = (acc >> (n - 8)) & 0xFF
                out.append(byte)
                if byte == 0xFF:
                    out.append(0)
                n -= 8
            acc &= (1 << n) - 1
        if n:
            byte = ((acc << (8 - n)) | ((1 << (8 - n)) - 1)) & 0xFF
            out.append(byte)
            if byte == 0xFF:
                out.append(0)
        bits.clear()
    pred = [0] * len(planes)
    nm = 0
    for my in range(mcuy):
        for mx in range(mcux):
            if restart and nm and nm % restart == 0:
                flush_bits()
                out += bytes([0xFF, 0xD0 + ((nm // restart - 1) & 7)])
                pred = [0] * len(planes)
            nm += 1
            for ci, (ch, cv) in enumerate(samp):
                for by in range(cv):
                    for bx in range(ch):
                        blk = blocks[ci][my * cv + by, mx * ch + bx]
                        t, v = mag(blk[0] - pred[ci])
                        pred[ci] = int(blk[0])
                        put(t, 4)
                        if t:
                            put(v, t)
                        run = 0
                        last = max([k for k in range(1, 64) if blk[_ZZ[k]] != 0], default=0)
                        for k in range(1, last + 1):
                            c = blk[_ZZ[k]]
                            if c == 0:
                                run += 1
                                continue
                            while run > 15:
                                put(ac_code[0xF0], 8)
                                run -= 16
                            t, v = mag(c)
                            put(ac_code[(run << 4) | t], 8)
                            put(v, t)
                            run = 0
                        if last < 63:
                            put(ac_code[0x00], 8)
    flush_bits()
    seg = lambda m, body: bytes([0xFF, m]) + struct.pack(">H", len(body) + 2) + body
    head = b"\xff\xd8" + seg(0xE0, b"JFIF\0\x01\x01\0\0\x01\0\x01\0\0")
    if orientation is not None:
        head += exif_app1(orientation)
    head += seg(0xDB, bytes([0]) + bytes(int(qt.reshape(-1)[_ZZ[k]]) for k in range(64)))
    nc = len(planes)
    head += seg(0xC0, struct.pack(">BHHB", 8, h, w, nc) + b"".join(bytes([ci + 1, (ch << 4) | cv, 0]) for ci, (ch, cv) in enumerate(samp)))
    head += seg(0xC4, bytes([0x00]) + bytes([0, 0, 0, 12] + [0] * 12) + bytes(range(12)))
    head += seg(0xC4, bytes([0x10]) + bytes([0] * 7 + [len(_AC_SYMS)] + [0] * 8) + bytes(_AC_SYMS))
    if restart:
        head += seg(0xDD, struct.pack(">H", restart))
    head += seg(0xDA, bytes([nc]) + b"".join(bytes([ci + 1, 0x00]) for ci in range(nc)) + bytes([0, 63, 0]))
    return head + bytes(out) + b"\xff\xd9"


# files Pillow cannot write (tests/golden/jpeg_cases.npz holds them next to CASES): name -> bytes. 4:4:0 is the layout of two of the
# reference's own data/demo files; the orientations are what cv2.imread applies
def extra_cases():
    return {
        "440-48x64": encode_custom(scene(48, 64, 1), 1, 2, q=6),
        "440-37x53-restart2": encode_custom(scene(37, 53, 2), 1, 2, q=10, restart=2),
        "440-2x9": encode_custom(scene(2, 9, 3), 1, 2, q=4),
        "440-31x18-orientation5": encode_custom(scene(31, 18, 4), 1, 2, q=8, orientation=5),
        "420-37x53-orientation6": with_exif_orientation(encode(scene(37, 53, 5), 90, 2), 6),
        "444-24x40-orientation3-MM": with_exif_orientation(encode(scene(24, 40, 6), 90, 0), 3, big_endian=True),
        "422-33x47-orientation8-progressive": with_exif_orientation(encode(scene(33, 47, 7), 85, 1, progressive=True), 8),
        "gray-20x30-orientation2": with_exif_orientation(encode(scene(20, 30, 8, gray=True), 80), 2),
        "420-21x40-orientation7": with_exif_orientation(encode(scene(21, 40, 9), 90, 2), 7),
        "420-40x21-orientation4": with_exif_orientation(encode(scene(40, 21, 10), 90, 2), 4),
    }


def cv2_like_bgr(data):
    """Pillow's decode turned by the EXIF orientation (ImageOps.exif_transpose), BGR: what cv2.imread returns for the file."""
    from PIL import ImageOps
    im = ImageOps.exif_transpose(Image.open(io.BytesIO(data)))
    return np.ascontiguousarray(np.asarray(im.convert("RGB"))[..., ::-1])
