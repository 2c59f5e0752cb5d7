"""CPU tests of the PNG decoder (cv2.imread in front of the hot path, reference ctpn/demo.py:59; the reference's demo
directory holds .jpg and .png). The decoder is host code by the nature of the format (csrc/png.cpp); the pin is Pillow's decode of the same
bytes in cv2.imread's channel order (cv2 is not in this image; both sit on zlib, and the colour-type handling restated here -- alpha
dropped, low bit depths scaled, palettes expanded -- is what Pillow's convert("RGB") and libpng's IMREAD_COLOR transforms agree on).
Nothing here needs a GPU or /root/reference.
"""
import io
import struct
import zlib

import numpy as np
import pytest
from PIL import Image

import ctpn_amd  # noqa: F401
from ctpn_amd import _binding as B
from util_jpeg import scene


def pillow_bgr(data):
    return np.ascontiguousarray(np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))[..., ::-1])


def save(im, **kw):
    buf = io.BytesIO()
    im.save(buf, "PNG", **kw)
    return buf.getvalue()


def chunk(kind, body):
    return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xFFFFFFFF)


def raw_png(w, h, depth, color, rows, interlace=0, plte=None, filters=None, idat_split=1, row_pass=None):
    """A PNG written by hand (Pillow cannot write every kind): rows = the packed scanlines (pass after pass when interlaced; row_pass[k] = the
    Adam7 pass of scanline k); filters = one filter type per scanline, applied here by the specification's forward formulas."""
    bits = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[color] * depth
    bpp = max(1, bits // 8)
    out = bytearray()
    for k, row in enumerate(rows):
        f = 0 if filters is None else filters[k]
        row = bytes(row)
        first = k == 0 or (row_pass is not None and row_pass[k] != row_pass[k - 1])       # the first scanline of a pass has no row above it
        prev = bytes(len(row)) if first else bytes(rows[k - 1])

        def a(i):
            return row[i - bpp] if i >= bpp else 0

        def c(i):
            return prev[i - bpp] if i >= bpp else 0

        def paeth(i):
            p = a(i) + prev[i] - c(i)
            pa, pb, pc = abs(p - a(i)), abs(p - prev[i]), abs(p - c(i))
            return a(i) if pa <= pb and pa <= pc else (prev[i] if pb <= pc else c(i))
        enc = {0: lambda i: row[i], 1: lambda i: row[i] - a(i), 2: lambda i: row[i] - prev[i], 3: lambda i: row[i] - ((a(i) + prev[i]) >> 1),
               4: lambda i: row[i] - paeth(i)}[f]
        out += bytes([f]) + bytes(enc(i) & 255 for i in range(len(row)))
    z = zlib.compress(bytes(out), 6)
    cut = [len(z) * i // idat_split for i in range(idat_split + 1)]
    data = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, color, 0, 0, interlace))
    if plte is not None:
        data += chunk(b"PLTE", bytes(plte))
    data += chunk(b"tEXt", b"Comment\0hand-made") + b"".join(chunk(b"IDAT", z[cut[i]:cut[i + 1]]) for i in range(idat_split)) + chunk(b"IEND", b"")
    return data


def pack_rows(pix, depth):
    """(h, w) integer samples of `depth` bits (or (h, w, c) 8-bit) -> packed scanlines."""
    if pix.ndim == 3:
        return [bytes(r.reshape(-1).astype(np.uint8)) for r in pix]
    rows = []
    for r in pix:
        bits = "".join(format(int(v), "0%db" % depth) for v in r)
        bits += "0" * (-len(bits) % 8)
        rows.append(bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8)))
    return rows


@pytest.fixture(params=["default", "zlib"], autouse=True)
def backend(request):
    """Every test runs twice: with the DEFLATE back end the library picks (libdeflate where the system has it) and with zlib forced."""
    B.png_backend(1 if request.param == "zlib" else 0)
    yield B.png_backend()
    B.png_backend(0)


def test_backends(backend, request):
    if "zlib" in request.node.name:
        assert backend == "zlib"
    else:
        import ctypes.util
        assert backend == ("libdeflate" if ctypes.util.find_library("deflate") else "zlib")


PIL_CASES = [
    ("rgb", lambda: Image.fromarray(scene(37, 53, 1)), {}),
    ("rgb-nocompress", lambda: Image.fromarray(scene(20, 31, 2)), {"compress_level": 0}),
    ("rgb-best", lambda: Image.fromarray(scene(64, 64, 3)), {"compress_level": 9, "optimize": True}),
    ("rgba", lambda: Image.fromarray(np.dstack([scene(33, 47, 4), scene(33, 47, 5, gray=True)])), {}),
    ("gray", lambda: Image.fromarray(scene(31, 42, 6, gray=True)), {}),
    ("gray-alpha", lambda: Image.fromarray(np.dstack([scene(18, 25, 7, gray=True), scene(18, 25, 8, gray=True)]), "LA"), {}),
    ("palette", lambda: Image.fromarray(scene(40, 56, 9)).quantize(200), {}),
    ("palette-16-colours", lambda: Image.fromarray(scene(23, 29, 10)).quantize(16), {"bits": 4}),
    ("palette-4-colours", lambda: Image.fromarray(scene(23, 30, 11)).quantize(4), {"bits": 2}),
    ("palette-2-colours", lambda: Image.fromarray(scene(17, 37, 12)).quantize(2), {"bits": 1}),
    ("palette-transparency", lambda: Image.fromarray(scene(20, 20, 13)).quantize(32), {"transparency": 3}),
    ("bilevel", lambda: Image.fromarray(scene(19, 43, 14, gray=True) > 128), {}),
    ("one-pixel", lambda: Image.fromarray(scene(1, 1, 15)), {}),
    ("one-column", lambda: Image.fromarray(scene(50, 1, 16)), {}),
    ("benchmark-size", lambda: Image.fromarray(scene(600, 900, 17)), {}),
]


@pytest.mark.parametrize("case", PIL_CASES, ids=lambda c: c[0])
def test_decode_equals_pillow_on_files_pillow_writes(case):
    _, make, kw = case
    data = save(make(), **kw)
    want = pillow_bgr(data)
    got = B.png_decode(data)
    assert got.shape == want.shape and np.array_equal(got, want)
    assert B.png_probe(data)[:2] == want.shape[:2]


@pytest.mark.parametrize("interlace", [0, 1], ids=["plain", "adam7"])
@pytest.mark.parametrize("kind", ["gray1", "gray2", "gray4", "gray8", "rgb8", "rgba8", "la8", "pal1", "pal2", "pal4", "pal8"])
def test_every_kind_filter_and_interlacing_on_hand_made_files(kind, interlace):
    """Files written here chunk by chunk: every colour type x bit depth the decoder takes, every filter type (one per scanline, cycling), IDAT
    split in three, Adam7 -- including sizes where some passes are empty."""
    color, depth = {"gray": 0, "rgb": 2, "rgba": 6, "la": 4, "pal": 3}[kind.rstrip("0123456789")], int(kind[-1])
    rng = np.random.default_rng(sum(kind.encode()) * 2 + interlace)
    for (h, w) in [(1, 1), (2, 3), (5, 9), (8, 8), (13, 21), (33, 17)]:
        ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[color]
        pix = rng.integers(0, 1 << depth, (h, w) if ch == 1 else (h, w, ch))
        plte = rng.integers(0, 256, 3 * (1 << depth)).astype(np.uint8) if color == 3 else None
        passes = [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)] if interlace else [(0, 0, 1, 1)]
        rows, which = [], []
        for p, (x0, y0, dx, dy) in enumerate(passes):
            sub = pix[y0::dy, x0::dx]
            if sub.shape[0] and sub.shape[1]:
                r = pack_rows(sub, depth)
                rows += r
                which += [p] * len(r)
        data = raw_png(w, h, depth, color, rows, interlace, plte, filters=[k % 5 for k in range(len(rows))], idat_split=3, row_pass=which)
        want = pillow_bgr(data)
        got = B.png_decode(data)
        assert np.array_equal(got, want), (kind, interlace, h, w)
        # and against the construction itself, independent of any decoder
        if color == 3:
            mine = plte.reshape(-1, 3)[pix][..., ::-1]
        elif ch == 1:
            mine = np.repeat((pix * (255 // ((1 << depth) - 1)))[..., None], 3, -1)
        elif ch == 2:
            mine = np.repeat(pix[..., :1], 3, -1)
        else:
            mine = pix[..., 2::-1]
        assert np.array_equal(got, mine.astype(np.uint8)), (kind, interlace, h, w)


def test_files_in_one_call_and_their_errors(tmp_path):
    names, want = [], []
    for i in range(7):
        data = save(Image.fromarray(scene(60, 90, 20 + i)), compress_level=i % 10)
        (tmp_path / ("p%d.png" % i)).write_bytes(data)
        names.append(str(tmp_path / ("p%d.png" % i)))
        want.append(pillow_bgr(data))
    for threads in (0, 1, 3):
        assert np.array_equal(B.decode_png_files(names, 60, 90, threads), np.stack(want))
    assert B.decode_png_files([], 60, 90).shape == (0, 60, 90, 3)
    (tmp_path / "other.png").write_bytes(save(Image.fromarray(scene(61, 90, 1))))
    (tmp_path / "deep.png").write_bytes(save(Image.fromarray((scene(60, 90, 2, gray=True).astype(np.uint16) * 257))))
    (tmp_path / "not.png").write_bytes(b"\xff\xd8 not a png")
    probe = B.png_probe_files(names[:2] + [str(tmp_path / n) for n in ("other.png", "deep.png", "not.png", "missing.png")], 2)
    assert probe.tolist() == [[60, 90, 2, 8], [60, 90, 2, 8], [61, 90, 2, 8], [0, 0, 0, 0], [0, 0, 0, 0], [0, 0, 0, 0]]
    for bad, code in (("other.png", -1), ("deep.png", B.CTPN_ERR_UNSUPPORTED), ("not.png", -1), ("missing.png", -1)):
        with pytest.raises(B.CtpnError) as e:
            B.decode_png_files(names[:2] + [str(tmp_path / bad)], 60, 90)
        assert e.value.code == code and bad in str(e.value)


def test_damaged_files_are_errors_not_crashes():
    data = save(Image.fromarray(scene(40, 56, 3)))
    rng = np.random.default_rng(0)
    with pytest.raises(B.CtpnError):
        B.png_decode(data[:40])
    with pytest.raises(B.CtpnError):
        B.png_decode(data[: len(data) // 2])                 # image data ends early
    idat = data.index(b"IDAT")
    flipped = bytearray(data)
    flipped[idat + 20] ^= 0x55
    with pytest.raises(B.CtpnError) as e:                    # the chunk's CRC catches it before inflate does
        B.png_decode(bytes(flipped))
    assert "CRC" in str(e.value)
    for _ in range(200):                                     # random damage anywhere: an error or an image, never a crash (ASan runs this too)
        junk = bytearray(data)
        for pos in rng.integers(8, len(data), 3):
            junk[pos] = int(rng.integers(0, 256))
        try:
            assert B.png_decode(bytes(junk)).shape == (40, 56, 3)
        except B.CtpnError as e:
            assert e.code in (-1, B.CTPN_ERR_UNSUPPORTED, -4)
    # valid CRCs around a corrupt zlib stream / a bad filter byte
    bad_filter = raw_png(4, 2, 8, 0, [bytes(4), bytes(4)], filters=None)
    z = zlib.compress(b"\x07" + bytes(4) + b"\x00" + bytes(4))
    bad_filter = bad_filter[: bad_filter.index(b"IDAT") - 4] + chunk(b"IDAT", z) + chunk(b"IEND", b"")
    with pytest.raises(B.CtpnError) as e:
        B.png_decode(bad_filter)
    assert "filter" in str(e.value)
    broken = bad_filter[: bad_filter.index(b"IDAT") - 4] + chunk(b"IDAT", b"\x78\x9c" + bytes(range(20))) + chunk(b"IEND", b"")
    with pytest.raises(B.CtpnError):
        B.png_decode(broken)


def test_batch_cli_routes_every_file_to_its_decoder(tmp_path, monkeypatch):
    """ctpn/demo_batch.py --decode gpu on a mixed directory with the device calls replaced by recorders (no GPU here; the GPU suite runs the
    real thing): JPEG files the library takes go to ctpn_decode_jpeg_files grouped by size and layout, PNG files to ctpn_decode_png_files
    (their pixels arrive at detect_submit byte-equal to Pillow's, resized where resize_im asks for it), everything else to Pillow; one result
    file per image."""
    import os
    from ctpn_amd.ctpn import demo_batch
    from util_jpeg import encode
    src, out = tmp_path / "in", tmp_path / "out"
    src.mkdir()
    (src / "a0.jpg").write_bytes(encode(scene(600, 900, 1), 90, 2))
    (src / "a1.jpg").write_bytes(encode(scene(600, 900, 2), 90, 2, progressive=True))
    (src / "a2.jpg").write_bytes(encode(scene(600, 900, 3), 90, 1))                                   # 4:2:2: its own batch
    Image.fromarray(scene(600, 900, 4)).convert("CMYK").save(str(src / "a3.jpg"), "JPEG")          # Pillow's
    Image.fromarray(scene(600, 900, 5)).save(str(src / "b0.png"))
    Image.fromarray(scene(600, 900, 6, gray=True)).save(str(src / "b1.png"))
    Image.fromarray(np.dstack([scene(300, 450, 7), scene(300, 450, 8, gray=True)])).save(str(src / "b2.png"))     # resize_im doubles it
    Image.fromarray(scene(600, 900, 9, gray=True).astype(np.uint16) * 257).save(str(src / "b3.png"))              # 16 bit: Pillow's
    calls = []

    class Ctx:
        def decode_jpeg_files(self, members, h, w, fx, fy):
            calls.append(("jpeg", [os.path.basename(m) for m in members], (h, w)))
            return 1, (len(members),) + tuple(B.resize_dims(h, w, fx, fy) if fx != 1.0 else (h, w))

        def detect_submit(self, images=None, slot=0, device_ptr=None, shape=None):
            calls.append(("submit", None if images is None else np.array(images), slot))
            self.n = getattr(self, "n", {})
            self.n[slot] = shape[0] if images is None else len(images)

        def detect_collect(self, slot, mode="H", line_capacity=512):
            return [np.zeros((0, 9))] * self.n[slot]

    class Net:
        ctx = Ctx()

        def ensure_capacity(self, n, h, w):
            calls.append(("capacity", n, h, w))

    monkeypatch.setattr(B, "resize_linear", lambda im, fx, fy, device_id=0: np.repeat(np.repeat(im, int(fy), -3), int(fx), -2))
    logs = []
    names = demo_batch.list_images(str(src))
    res = demo_batch.run(Net(), names, str(out), batch=4, write_images=False, log=logs.append, decode="gpu")
    assert sorted(res) == sorted(names) and len(os.listdir(str(out))) == 8
    assert "3 decoded on the device, 3 PNG files by the library, 2 on the host" in logs[0], logs
    jpeg = [c for c in calls if c[0] == "jpeg"]
    assert sorted(c[1] for c in jpeg) == [["a0.jpg", "a1.jpg"], ["a2.jpg"]]
    host_batches = [c[1] for c in calls if c[0] == "submit" and c[1] is not None]
    want = {n: pillow_bgr(open(str(src / n), "rb").read()) for n in ("b0.png", "b1.png", "b2.png")}
    want["b2.png"] = np.repeat(np.repeat(want["b2.png"], 2, 0), 2, 1)
    flat = [im for b in host_batches for im in b]
    for n, w in want.items():
        assert any(im.shape == w.shape and np.array_equal(im, w) for im in flat), n


def test_sixteen_bit_files_keep_their_high_byte_on_the_host_path(tmp_path):
    """16-bit PNG files are not the library's (CTPN_ERR_UNSUPPORTED); lib/utils/image.py's imread gives what libpng hands cv2.imread(IMREAD_COLOR)
    -- the high byte of every sample (png_set_strip_16) -- for every colour type, including 16-bit gray, where Pillow's own convert("RGB")
    clips at 255 instead."""
    from ctpn_amd.lib.utils import image as imutil
    rng = np.random.default_rng(0)
    for color, ch in ((0, 1), (2, 3), (4, 2), (6, 4)):
        v = rng.integers(0, 65536, (7, 9, ch))
        rows = [b"".join(struct.pack(">H", int(x)) for x in r.reshape(-1)) for r in v]
        data = raw_png(9, 7, 16, color, rows)
        with pytest.raises(B.CtpnError) as e:
            B.png_decode(data)
        assert e.value.code == B.CTPN_ERR_UNSUPPORTED
        (tmp_path / "deep.png").write_bytes(data)
        hi = (v >> 8).astype(np.uint8)
        want = np.repeat(hi[..., :1], 3, -1) if ch in (1, 2) else hi[..., 2::-1]
        assert np.array_equal(imutil.imread(str(tmp_path / "deep.png")), want), color


def test_absurd_sizes_are_errors_not_allocations():
    """A header that announces 65535 x 65535 pixels (12 GB of scanlines) over a few bytes of data: refused before anything is allocated for it."""
    z = zlib.compress(b"\0" * 64)
    data = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 65535, 65535, 8, 2, 0, 0, 0)) + chunk(b"IDAT", z) + chunk(b"IEND", b"")
    assert B.png_probe(data)[:2] == (65535, 65535)
    out = np.zeros((16,), np.uint8)
    lib = B.load_library()
    keep, ptr, n = B._bytes_ptr(data)
    assert lib.ctpn_png_decode(ptr, n, out.ctypes.data_as(__import__("ctypes").POINTER(__import__("ctypes").c_uint8)), out.size) == -4      # capacity
    big = np.zeros((1,), np.uint8)
    assert lib.ctpn_png_decode(ptr, n, big.ctypes.data_as(__import__("ctypes").POINTER(__import__("ctypes").c_uint8)), 1 << 40) == B.CTPN_ERR_UNSUPPORTED
    assert b"too large" in lib.ctpn_last_error()


def test_imread_takes_png_files_through_the_library_and_agrees_with_pillow(tmp_path):
    """lib/utils/image.py's imread (the cv2.imread of ctpn/demo.py:59): PNG through ctpn_png_decode, the rest -- and what the library refuses --
    through Pillow; the same pixels either way."""
    from ctpn_amd.lib.utils import image as imutil
    for name, make, kw in PIL_CASES[:12]:
        p = tmp_path / (name + ".png")
        p.write_bytes(save(make(), **kw))
        assert np.array_equal(imutil.imread(str(p)), pillow_bgr(p.read_bytes())), name
    (tmp_path / "broken.png").write_bytes(b"\x89PNG\r\n\x1a\n" + bytes(40))
    with pytest.raises(Exception):
        imutil.imread(str(tmp_path / "broken.png"))           # the library refuses it, then Pillow does


def test_batch_cli_refuses_edited_pixel_means(tmp_path):
    """cfg.PIXEL_MEANS is read at run time by the reference (lib/fast_rcnn/test.py:7-11); the uint8 batch feed has the means compiled into its
    first kernel, so an edited value is an error there, not a silent no-op."""
    from ctpn_amd.ctpn import demo_batch
    from ctpn_amd.lib.fast_rcnn.config import cfg
    keep = cfg.PIXEL_MEANS
    cfg.PIXEL_MEANS = np.array([[[100.0, 110.0, 120.0]]])
    try:
        with pytest.raises(ValueError) as e:
            demo_batch.run(None, [], str(tmp_path), decode="gpu")
        assert "PIXEL_MEANS" in str(e.value)
    finally:
        cfg.PIXEL_MEANS = keep
    demo_batch._check_uint8_feed_config()
    keep = cfg.TEST.RPN_POST_NMS_TOP_N
    cfg.TEST.RPN_POST_NMS_TOP_N = 300
    try:
        with pytest.raises(ValueError) as e:
            demo_batch.run(None, [], str(tmp_path))
        assert "RPN_POST_NMS_TOP_N" in str(e.value)
    finally:
        cfg.TEST.RPN_POST_NMS_TOP_N = keep
    demo_batch._check_uint8_feed_config()
