"""CPU tests of the JPEG decoder's seams (cv2.imread in front of the hot path, reference ctpn/demo.py:59).

  Pillow (libjpeg-turbo; the pin)  ==  oracle/jpeg_ref.py, whole pipeline in numpy                              (oracle pinned)
  library host half (ctpn_jpeg_entropy_decode, C++)  ==  oracle entropy decoder, coefficient for coefficient       (bit-exact)
  library host half -> oracle pixel half  ==  Pillow                                                             (bit-exact)
  library host half -> the device half's per-sample source (csrc/jpeg_pixel.h) compiled for the host  ==  Pillow  (bit-exact)

The device half (IDCT / upsampling / colour kernels) is the GPU suite's: tests/test_gpu_jpeg.py. Nothing here needs a GPU or
/root/reference.
"""
import io
import os

import numpy as np
import pytest

import ctpn_amd  # noqa: F401
from ctpn_amd import _binding as B
from oracle import jpeg_ref as J
from util_jpeg import CASES, case_id, cv2_like_bgr, encode, encode_custom, extra_cases, pillow_bgr, scene, with_exif_orientation, with_luma_sampling


def test_committed_vectors(golden_dir):
    """tests/golden/jpeg_cases.npz (oracle/make_jpeg_golden.py: files + libjpeg-turbo's decode of them): the oracle, the library's host
    half in front of the oracle's pixel half, and the Pillow installed here all reproduce the committed pixels."""
    import os
    g = np.load(os.path.join(golden_dir, "jpeg_cases.npz"))
    assert len(g["names"]) == len(CASES) + len(extra_cases())
    for name in g["names"]:
        data, want = g["file_" + name].tobytes(), g["bgr_" + name]
        assert np.array_equal(J.imread_bgr(data), want), name
        planes, qt, lay = B.jpeg_entropy_decode(data)
        got = J.pixels_from_coefficients(planes, [qt[c] for c in range(lay["ncomp"])], lay["h"], lay["w"], lay["hs"], lay["vs"])
        assert np.array_equal(J.apply_orientation(got, lay["orientation"]), want), name
        assert np.array_equal(cv2_like_bgr(data), want), name
    for name, data in extra_cases().items():                  # the encoder of tests/util_jpeg.py still writes the committed bytes
        assert g["file_" + name].tobytes() == data, name


@pytest.mark.parametrize("case", CASES, ids=case_id)
def test_oracle_equals_pillow(case):
    h, w, q, sub, gray, kw = case
    data = encode(scene(h, w, h + w, gray), q, sub, **kw)
    want = pillow_bgr(data)
    got = J.imread_bgr(data)
    assert got.shape == want.shape == (h, w, 3)
    assert np.array_equal(got, want), int(np.abs(got.astype(int) - want.astype(int)).max())


@pytest.mark.parametrize("case", CASES, ids=case_id)
def test_library_host_half_equals_the_oracles_coefficients(case):
    h, w, q, sub, gray, kw = case
    data = encode(scene(h, w, h + w, gray), q, sub, **kw)
    frame, want = J.coefficients(data)
    planes, qt, lay = B.jpeg_entropy_decode(data)
    assert (lay["h"], lay["w"], lay["ncomp"]) == (h, w, 1 if gray else 3)
    assert (lay["hs"], lay["vs"]) == ((1, 1) if gray or sub == 0 else ((2, 1) if sub == 1 else (2, 2)))
    assert len(planes) == len(want)
    for c, (g, o) in enumerate(zip(planes, want)):
        assert g.shape == o.shape, (c, g.shape, o.shape)
        assert np.array_equal(g.astype(np.int32), o), c
    for c, comp in enumerate(frame["comps"]):
        assert np.array_equal(qt[c].astype(np.int64), frame["qt"][comp[3]])


# the sizes the hot path is quoted on (BASELINE.json configs[1]) and its neighbours: the library's host half in front of the vectorised oracle
@pytest.mark.parametrize("progressive", [False, True], ids=["sequential", "progressive"])
@pytest.mark.parametrize("geom", [(600, 900, 90, 2), (600, 900, 75, 0), (601, 899, 95, 2), (255, 1201, 60, 2), (600, 900, 85, 1), (599, 901, 85, 1)],
                         ids=lambda g: "%dx%d-q%d-s%d" % g)
def test_host_half_then_oracle_pixels_equals_pillow_at_full_size(geom, progressive):
    h, w, q, sub = geom
    data = encode(scene(h, w, 5), q, sub, progressive=progressive)
    assert (b"\xff\xc2" in data) == progressive
    planes, qt, lay = B.jpeg_entropy_decode(data)
    got = J.pixels_from_coefficients(planes, [qt[c] for c in range(lay["ncomp"])], h, w, lay["hs"], lay["vs"])
    want = pillow_bgr(data)
    assert np.array_equal(got, want)


def test_random_sizes_qualities_and_layouts_equal_pillow():
    """A hundred and twenty files of random size (every partial-MCU case, the narrow images whose chroma libjpeg replicates), quality 1..100 (quantisation
    tables from all-255 to all-1), layout, optimised tables, restart intervals, every fourth one progressive."""
    rng = np.random.default_rng(7)
    for k in range(120):
        h, w = int(rng.integers(1, 120)), int(rng.integers(1, 120))
        q, sub, gray = int(rng.integers(1, 101)), int(rng.choice([0, 1, 2])), bool(rng.integers(0, 5) == 0)
        kw = {"optimize": True} if k % 3 == 0 else ({"restart_marker_blocks": int(rng.integers(1, 9))} if k % 3 == 1 else {})
        if k % 4 == 3:
            kw["progressive"] = True
        data = encode(scene(h, w, k, gray), q, sub, **kw)
        planes, qt, lay = B.jpeg_entropy_decode(data)
        got = J.pixels_from_coefficients(planes, [qt[c] for c in range(lay["ncomp"])], h, w, lay["hs"], lay["vs"])
        assert np.array_equal(got, pillow_bgr(data)), (k, h, w, q, sub, gray, kw)


@pytest.fixture(scope="module")
def device_source_on_host(root, tmp_path_factory):
    """csrc/jpeg_pixel.h -- the text jpeg_idct_kernel and jpeg_color_kernel are made of -- compiled with g++ (tests/jpeg_pixel_host.cpp)."""
    import ctypes as C
    import subprocess
    so = str(tmp_path_factory.mktemp("jpeg_pixel_host") / "libjpeg_pixel_host.so")
    subprocess.run(["g++", "-O2", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-o", so, os.path.join(root, "tests", "jpeg_pixel_host.cpp")], check=True)
    lib, L = C.CDLL(so), B.load_library()

    def decode(data):
        h, w, nc, hs = B.jpeg_probe(data)
        cap = int(L.ctpn_jpeg_coef_capacity(h, w))
        coef, qt, l8 = np.zeros(cap, np.int16), np.zeros((3, 64), np.uint16), np.zeros(8, np.int32)
        keep, ptr, n = B._bytes_ptr(data)
        B._check(L.ctpn_jpeg_entropy_decode(ptr, n, coef.ctypes.data_as(C.POINTER(C.c_int16)), cap, qt.ctypes.data_as(C.POINTER(C.c_uint16)),
                                            l8.ctypes.data_as(C.POINTER(C.c_int))))
        out = np.zeros((h, w, 3), np.uint8)               # h, w: the probe's, i.e. the TURNED image's (EXIF orientation)
        lib.jpeg_pixels_host(coef.ctypes.data_as(C.c_void_p), qt.ctypes.data_as(C.c_void_p), l8.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        return out
    return decode


def test_the_device_halfs_source_compiled_for_the_host_equals_pillow(device_source_on_host):
    """The per-sample arithmetic of the two kernels (IDCT passes, the three chroma layouts' upsampling, colour conversion), from the very
    text hipcc compiles: every fixed case, then two hundred random files over all layouts, sequential and progressive."""
    for case in CASES:
        h, w, q, sub, gray, kw = case
        data = encode(scene(h, w, h + w, gray), q, sub, **kw)
        assert np.array_equal(device_source_on_host(data), pillow_bgr(data)), case_id(case)
    rng = np.random.default_rng(9)
    for k in range(200):
        h, w = int(rng.integers(1, 100)), int(rng.integers(1, 100))
        q, sub, gray = int(rng.integers(1, 101)), int(rng.choice([0, 1, 2])), bool(rng.integers(0, 6) == 0)
        data = encode(scene(h, w, k, gray), q, sub, progressive=bool(k % 2))
        assert np.array_equal(device_source_on_host(data), pillow_bgr(data)), (k, h, w, q, sub, gray)
    # 4:4:0 (luma 1 x 2: h1v2 upsampling; files of tests/util_jpeg.py's own encoder, Pillow's decode is the pin) and the eight EXIF
    # orientations on every layout (the colour kernel's index map)
    for k in range(60):
        h, w = int(rng.integers(1, 80)), int(rng.integers(1, 80))
        data = encode_custom(scene(h, w, k), 1, 2, q=int(rng.integers(2, 30)), restart=int(rng.integers(0, 4)))
        assert np.array_equal(device_source_on_host(data), pillow_bgr(data)), ("440", k, h, w)
    for k in range(64):
        h, w, o = int(rng.integers(1, 70)), int(rng.integers(1, 70)), 1 + k % 8
        sub = (k // 8) % 4
        data = encode_custom(scene(h, w, k), 1, 2, orientation=o) if sub == 3 else with_exif_orientation(encode(scene(h, w, k), 85, sub, progressive=bool(k & 16)), o, bool(k & 32))
        want = cv2_like_bgr(data)
        assert want.shape[:2] == ((w, h) if o >= 5 else (h, w))
        assert np.array_equal(device_source_on_host(data), want), ("orientation", o, sub, h, w)


def test_the_references_own_demo_files(golden_dir, device_source_on_host):
    """tests/golden/demo_files.npz (oracle/make_demo_golden.py): data/demo/00{6,7,8,9}.jpg + 010.png of the reference tree, with the SHA-256
    of what cv2.imread returns for each. CPU halves: the header scan takes all four JPEG files (006 / 009: 4:4:0; 008: EXIF orientation 6);
    host half + the device half's source compiled for the host reproduce the committed pixels; so does the oracle's pixel half; the PNG goes
    through the library's PNG decoder; and the Pillow installed here still agrees with the committed hashes."""
    import hashlib
    g = np.load(os.path.join(golden_dir, "demo_files.npz"))
    layouts = {}
    for nm in g["names"]:
        key = str(nm).replace(".", "_")
        data, want_sha, shape = g["file_" + key].tobytes(), str(g["sha256_" + key]), tuple(g["shape_" + key])
        if str(nm).endswith(".png"):
            got = B.png_decode(data)
        else:
            pr = B.jpeg_probe(data)
            assert pr[:2] == shape[:2]
            layouts[str(nm)] = (pr[3] & 0xff, (pr[3] >> 8) + 1)
            got = device_source_on_host(data)
            planes, qt, lay = B.jpeg_entropy_decode(data)
            ora = J.apply_orientation(J.pixels_from_coefficients(planes, [qt[c] for c in range(3)], lay["h"], lay["w"], lay["hs"], lay["vs"]), lay["orientation"])
            assert hashlib.sha256(np.ascontiguousarray(ora).tobytes()).hexdigest() == want_sha, nm
            assert hashlib.sha256(cv2_like_bgr(data).tobytes()).hexdigest() == want_sha, nm
        assert got.shape == shape and np.array_equal(got[:32, :32], g["windows_" + key][0]), nm
        assert hashlib.sha256(got.tobytes()).hexdigest() == want_sha, nm
    assert layouts == {"006.jpg": (0x12, 1), "007.jpg": (2, 1), "008.jpg": (1, 6), "009.jpg": (0x12, 1)}


def test_probe_reads_the_header_only():
    data = encode(scene(37, 53, 1), 90, 2)
    assert B.jpeg_probe(data) == (37, 53, 3, 2)
    assert B.jpeg_probe(data[: data.index(b"\xff\xda") + 14])[:2] == (37, 53)      # everything up to the scan header is enough
    assert B.jpeg_probe(encode(scene(20, 30, 1, gray=True))) == (20, 30, 1, 1)
    assert B.jpeg_probe(encode(scene(20, 30, 1), 90, 0)) == (20, 30, 3, 1)
    assert B.jpeg_probe(encode(scene(20, 30, 1), 90, 1)) == (20, 30, 3, 0x21)               # 4:2:2: 2 horizontally, 1 vertically
    assert B.jpeg_probe(encode(scene(20, 30, 1), 90, 2, progressive=True)) == (20, 30, 3, 2)


def test_probe_files_scans_a_directory_in_one_call(tmp_path):
    """ctpn_jpeg_probe_files: per-file outcomes are data (h = 0: not for the device decoder), never an error of the call."""
    from PIL import Image
    good = encode(scene(37, 53, 1), 90, 2)
    files = {
        "a.jpg": good,
        "b.jpg": encode(scene(20, 30, 2), 80, 0),
        "c.jpg": encode(scene(24, 40, 3, gray=True), 80),
        "d.jpg": encode(scene(40, 56, 4), 90, 2, progressive=True),                      # progressive: the host half's business alone
        "h.jpg": with_luma_sampling(encode(scene(40, 56, 4), 90, 2), 0x41),              # 4:1:1: unsupported kind
        "j.jpg": encode_custom(scene(40, 56, 4), 1, 2),                                  # 4:4:0
        "k.jpg": with_exif_orientation(encode(scene(40, 56, 4), 90, 2), 6),              # stored 40 x 56, shown 56 x 40
        "i.jpg": encode(scene(40, 56, 4), 90, 1),                                        # 4:2:2
        "e.jpg": b"not a jpeg at all",
        # 150 KB of APP1 segments in front of the frame header: more than the 64 KB the scan reads first
        "f.jpg": good[:2] + b"".join(b"\xff\xe1" + (50002).to_bytes(2, "big") + bytes(50000) for _ in range(3)) + good[2:],
    }
    for k, v in files.items():
        (tmp_path / k).write_bytes(v)
    Image.fromarray(scene(16, 16, 5)).save(str(tmp_path / "g.png"))
    names = [str(tmp_path / k) for k in ("a.jpg", "b.jpg", "c.jpg", "d.jpg", "e.jpg", "f.jpg", "g.png", "missing.jpg", "h.jpg", "i.jpg", "j.jpg", "k.jpg")]
    for threads in (0, 1, 3):
        got = B.jpeg_probe_files(names, threads)
        assert got.tolist() == [[37, 53, 3, 2], [20, 30, 3, 1], [24, 40, 1, 1], [40, 56, 3, 2], [0, 0, 0, 0], [37, 53, 3, 2], [0, 0, 0, 0], [0, 0, 0, 0], [0, 0, 0, 0], [40, 56, 3, 0x21],
                                [40, 56, 3, 0x12], [56, 40, 3, 2 | (5 << 8)]]
    assert B.jpeg_probe_files([]).shape == (0, 4)
    # and the file with the long header decodes like the plain one (Pillow agrees)
    planes, qt, lay = B.jpeg_entropy_decode(files["f.jpg"])
    assert np.array_equal(J.pixels_from_coefficients(planes, [qt[c] for c in range(3)], 37, 53, 2), pillow_bgr(files["f.jpg"]))


@pytest.mark.parametrize("hv, progressive", [(0x14, False), (0x41, False), (0x22 + 0x20, True)], ids=["1x4", "411", "4x2-progressive"])
def test_files_of_other_kinds_are_reported_as_unsupported_not_decoded_wrongly(hv, progressive):
    data = with_luma_sampling(encode(scene(40, 56, 2), 90, 2, progressive=progressive), hv)
    with pytest.raises(B.CtpnError) as e:
        B.jpeg_probe(data)
    assert e.value.code == B.CTPN_ERR_UNSUPPORTED
    with pytest.raises(B.CtpnError) as e:
        B.jpeg_entropy_decode(data)
    assert e.value.code == B.CTPN_ERR_UNSUPPORTED


def test_rgb_coded_files_are_unsupported_not_decoded_as_ycbcr():
    """libjpeg's colour-space rule (JFIF marker, else the Adobe marker's transform flag, else the component ids): a three-component file that
    stores RGB must not go through the YCbCr conversion. Pillow writes such files with keep_rgb (Adobe transform 0, ids 'R' 'G' 'B')."""
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(scene(40, 56, 1)).save(buf, "JPEG", quality=90, keep_rgb=True)
    data = buf.getvalue()
    assert b"Adobe" in data[:40]
    for fn in (B.jpeg_probe, B.jpeg_entropy_decode):
        with pytest.raises(B.CtpnError) as e:
            fn(data)
        assert e.value.code == B.CTPN_ERR_UNSUPPORTED and "RGB" in str(e.value)
    with pytest.raises(J.Unsupported):
        J.imread_bgr(data)
    # the same entropy data under a JFIF header is YCbCr by the rule (libjpeg would convert it, so does this decoder): only the markers decide
    i = data.index(b"\xff\xee")
    seg = 2 + int.from_bytes(data[i + 2:i + 4], "big")
    jfif = data[:i] + b"\xff\xe0\x00\x10JFIF\x00\x01\x01\x00\x00\x01\x00\x01\x00\x00" + data[i + seg:]
    assert B.jpeg_probe(jfif)[:3] == (40, 56, 3)
    assert np.array_equal(pillow_bgr(jfif), J.imread_bgr(jfif))
    # an Adobe marker with transform 1 (YCbCr), no JFIF: taken
    ycc = data[:i + 15] + b"\x01" + data[i + 16:]
    assert B.jpeg_probe(ycc)[:3] == (40, 56, 3)
    assert np.array_equal(pillow_bgr(ycc), J.imread_bgr(ycc))


def test_cmyk_is_unsupported():
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(scene(24, 24, 3)).convert("CMYK").save(buf, "JPEG")
    with pytest.raises(B.CtpnError) as e:
        B.jpeg_probe(buf.getvalue())
    assert e.value.code == B.CTPN_ERR_UNSUPPORTED


def test_damaged_files_are_errors_not_crashes():
    data = encode(scene(64, 64, 4), 90, 2)
    with pytest.raises(B.CtpnError):
        B.jpeg_probe(b"\x89PNG\r\n\x1a\n" + data)
    with pytest.raises(B.CtpnError):
        B.jpeg_probe(data[:20])
    # cut inside the entropy-coded segment: libjpeg warns, pretends the missing data is zero bits and leaves the MCUs it never reaches
    # empty; codes decoded from the padding are something else, so such a file is refused and goes to the host decoder -- no
    # crash, no read past the end (the ASan build runs this too)
    with pytest.raises(B.CtpnError) as e:
        B.jpeg_entropy_decode(data[: len(data) // 2])
    assert e.value.code == B.CTPN_ERR_UNSUPPORTED and "truncated" in str(e.value)
    # random bytes after a valid header: either an error or garbage coefficients, never a crash
    rng = np.random.default_rng(0)
    head = data[: data.index(b"\xff\xda") + 14]
    for _ in range(20):
        junk = head + rng.integers(0, 256, 600, dtype=np.uint8).tobytes()
        try:
            B.jpeg_entropy_decode(junk)
        except B.CtpnError as e:
            assert e.code in (-1, B.CTPN_ERR_UNSUPPORTED)


def test_damaged_progressive_files_are_errors_not_crashes():
    """A progressive file cut anywhere (inside a table, a scan header, any of its ten scans) or with flipped bytes: an error, or what the
    scans so far delivered (libjpeg's behaviour, with a warning) -- never a crash or a read past the end (the ASan build runs this too)."""
    data = encode(scene(40, 56, 4), 85, 2, progressive=True, restart_marker_blocks=2)
    first_scan = data.index(b"\xff\xda")
    rng = np.random.default_rng(1)
    for cut in sorted(set(rng.integers(first_scan, len(data), 60).tolist()) | {first_scan + 2, first_scan + 13, len(data) - 1, len(data) - 2}):
        try:
            planes, qt, lay = B.jpeg_entropy_decode(data[:cut])
            assert planes[0].shape == (6, 8, 64)
        except B.CtpnError as e:
            assert e.code in (-1, B.CTPN_ERR_UNSUPPORTED)
    for _ in range(200):
        junk = bytearray(data)
        for pos in rng.integers(first_scan, len(data), 4):
            junk[pos] = int(rng.integers(0, 256))
        try:
            B.jpeg_entropy_decode(bytes(junk))
        except B.CtpnError as e:
            assert e.code in (-1, B.CTPN_ERR_UNSUPPORTED)


def test_progressive_scan_kinds_are_all_present_in_the_test_files():
    """The four scan kinds of jdphuff.c -- DC first / refinement, AC first / refinement -- occur in what Pillow writes (libjpeg's
    jpeg_simple_progression), so the byte-equality above covers all of them."""
    data = encode(scene(48, 64, 1), 90, 2, progressive=True)
    kinds, i = set(), 2
    while i + 4 <= len(data):
        if data[i] != 0xFF or data[i + 1] in (0x00, 0xFF) or 0xD0 <= data[i + 1] <= 0xD9:
            i += 1
            continue
        m, L = data[i + 1], int.from_bytes(data[i + 2:i + 4], "big")
        if m == 0xDA:
            ns = data[i + 4]
            ss, ahl = data[i + 5 + 2 * ns], data[i + 7 + 2 * ns]
            kinds.add(("dc" if ss == 0 else "ac", "first" if ahl >> 4 == 0 else "refine"))
        i += 2 + L
    assert kinds == {("dc", "first"), ("dc", "refine"), ("ac", "first"), ("ac", "refine")}


def test_coefficient_capacity_covers_every_supported_layout():
    lib = B.load_library()
    for (h, w) in [(1, 1), (8, 8), (9, 9), (600, 900), (601, 899), (17, 1201)]:
        cap = lib.ctpn_jpeg_coef_capacity(h, w)
        for sub, gray in [(0, False), (1, False), (2, False), (2, True)]:
            planes, _, _ = B.jpeg_entropy_decode(encode(scene(h, w, 1, gray), 50, sub))
            assert sum(p.size for p in planes) <= cap
    assert lib.ctpn_jpeg_coef_capacity(0, 10) == 0


@pytest.mark.parametrize("big_endian", [False, True], ids=["II", "MM"])
def test_exif_orientation_is_applied_like_cv2_imread_applies_it(tmp_path, big_endian):
    """cv2.imread turns a JPEG by its EXIF orientation (OpenCV >= 3.1, default flags: what ctpn/demo.py:59 calls). The device decoder applies
    the tag in its colour kernel (the probe reports the TURNED size and the orientation), lib/utils/image.py's imread applies it for the
    files that go through Pillow: the eight orientations against their numpy statement, on both paths' CPU halves."""
    from ctpn_amd.lib.utils import image as imutil
    from ctpn_amd.ctpn import demo_batch
    plain = encode(scene(24, 40, 3), 95, 0)
    base = pillow_bgr(plain)
    turned = {1: base, 2: base[:, ::-1], 3: base[::-1, ::-1], 4: base[::-1], 5: base.transpose(1, 0, 2), 6: np.rot90(base, -1),
              7: base[::-1, ::-1].transpose(1, 0, 2), 8: np.rot90(base, 1)}
    names = []
    for o in range(1, 9):
        data = with_exif_orientation(plain, o, big_endian)
        p = tmp_path / ("o%d.jpg" % o)
        p.write_bytes(data)
        names.append(str(p))
        assert B.jpeg_probe(data) == turned[o].shape[:2] + (3, 1 | ((o - 1) << 8))
        assert B.jpeg_entropy_decode(data)[2]["orientation"] == o
        assert np.array_equal(J.imread_bgr(data), turned[o]) and np.array_equal(cv2_like_bgr(data), turned[o])
        got = imutil.imread(str(p))
        assert got.shape == turned[o].shape and np.array_equal(got, turned[o]), o
        assert demo_batch.image_size(str(p)) == turned[o].shape[:2]
    assert B.jpeg_probe_files(names)[:, :2].tolist() == [list(turned[o].shape[:2]) for o in range(1, 9)]
    # two EXIF segments: the first decides, whatever the second says (a scan that went on to later APP1 segments turned a
    # file that cv2.imread / Pillow do not turn); an APP1 that is not EXIF (XMP) in front of the EXIF one is skipped
    from util_jpeg import exif_app1
    j = 4 + ((plain[4] << 8) | plain[5]) if plain[2:4] == b"\xff\xe0" else 2
    xmp = b"http://ns.adobe.com/xap/1.0/\0<x/>"
    xmp = b"\xff\xe1" + (len(xmp) + 2).to_bytes(2, "big") + xmp
    for first, second, want_o in ((1, 6, 1), (6, 3, 6), (8, 1, 8)):
        data = plain[:j] + exif_app1(first, big_endian) + exif_app1(second, not big_endian) + plain[j:]
        assert B.jpeg_probe(data) == turned[want_o].shape[:2] + (3, 1 | ((want_o - 1) << 8)), (first, second)
        assert np.array_equal(J.imread_bgr(data), turned[want_o]) and np.array_equal(cv2_like_bgr(data), turned[want_o]), (first, second)
    data = plain[:j] + xmp + exif_app1(5, big_endian) + plain[j:]
    assert B.jpeg_probe(data) == turned[5].shape[:2] + (3, 1 | (4 << 8))
    assert np.array_equal(J.imread_bgr(data), turned[5]) and np.array_equal(cv2_like_bgr(data), turned[5])
    # a damaged Exif segment is no orientation, not a crash (ASan runs this): truncated IFD, offsets past the segment
    for cut in range(8, 30, 3):
        body = b"Exif\0\0" + (b"MM" if big_endian else b"II") + bytes(range(cut))
        junk = plain[:2] + b"\xff\xe1" + (len(body) + 2).to_bytes(2, "big") + body + plain[2:]
        assert B.jpeg_probe(junk)[:2] == (24, 40)


def test_440_files_host_half_and_oracle_equal_pillow():
    """4:4:0 (luma sampled 1 x 2: two of the reference's own data/demo files are): the library's host half against the oracle's entropy
    decoder, and both pixel paths against Pillow, whole and partial MCUs, one-pixel and two-row images, restart intervals."""
    for k, (h, w, rst) in enumerate([(48, 64, 0), (37, 53, 0), (1, 1, 0), (2, 9, 0), (3, 40, 0), (20, 2, 3), (33, 17, 2), (16, 8, 1)]):
        data = encode_custom(scene(h, w, 10 + k), 1, 2, q=4 + 3 * k, restart=rst)
        assert B.jpeg_probe(data) == (h, w, 3, 0x12)
        planes, qt, lay = B.jpeg_entropy_decode(data)
        assert (lay["hs"], lay["vs"], lay["orientation"]) == (1, 2, 1)
        f, blocks = J.coefficients(data)
        for a, b in zip(planes, blocks):
            assert np.array_equal(a, b)
        want = pillow_bgr(data)
        assert np.array_equal(J.imread_bgr(data), want)
        assert np.array_equal(J.pixels_from_coefficients(planes, [qt[c] for c in range(3)], h, w, 1, 2), want)


def _scan_offsets(data):
    """Offsets of the SOS markers and of EOI."""
    out, i = [], 2
    while i + 4 <= len(data):
        if data[i] != 0xFF or data[i + 1] in (0x00, 0xFF) or 0xD0 <= data[i + 1] <= 0xD8:
            i += 1
            continue
        if data[i + 1] == 0xD9:
            out.append(i)
            break
        if data[i + 1] == 0xDA:
            out.append(i)
        i += 2 + int.from_bytes(data[i + 2:i + 4], "big")
    return out


def test_incomplete_files_are_refused_not_decoded_differently_from_libjpeg():
    """libjpeg smooths a progressive image whose last scans are missing (jdcoefct.c) and has its own rules for entropy
    data that ends early; the plain IDCT of the coefficients delivered so far is NOT what cv2 / Pillow return for such a file. They are
    CTPN_ERR_UNSUPPORTED (h = 0 in the directory scan's terms: decode fails, the caller's host decoder takes the file) -- checked here
    together with the fact that the refused files really do differ, i.e. that refusing is necessary."""
    from PIL import Image, ImageFile
    img = scene(64, 80, 3)
    prog = encode(img, 85, 2, progressive=True)
    offs = _scan_offsets(prog)
    assert len(offs) >= 8                                   # libjpeg's default script: ten scans + EOI
    complete = B.jpeg_entropy_decode(prog)
    for cut in offs[1:-1]:                                   # a legal file: the first k scans, then EOI
        part = prog[:cut] + b"\xff\xd9"
        with pytest.raises(B.CtpnError) as e:
            B.jpeg_entropy_decode(part)
        assert e.value.code == B.CTPN_ERR_UNSUPPORTED and "last scans" in str(e.value)
    # ... and libjpeg's picture of the DC-only file is indeed not the IDCT of its coefficients (block smoothing)
    dc_only = prog[:offs[1]] + b"\xff\xd9"
    f, blocks = J.coefficients(dc_only)
    plain_idct = J.pixels_from_coefficients(blocks, [f["qt"][c[3]] for c in f["comps"]], 64, 80, 2, 2)
    assert np.abs(plain_idct.astype(int) - pillow_bgr(dc_only)).max() > 20
    # truncated inside a scan: sequential and progressive, with and without restart markers
    for data in (encode(img, 85, 2), encode(img, 85, 0, restart_marker_blocks=2), prog, encode(img, 85, 1, progressive=True, restart_marker_blocks=3)):
        first = data.index(b"\xff\xda")
        rng = np.random.default_rng(len(data))
        refused = 0
        for cut in rng.integers(first + 20, len(data) - 2, 25).tolist():
            try:
                B.jpeg_entropy_decode(data[:cut])
            except B.CtpnError as e:
                assert e.code in (-1, B.CTPN_ERR_UNSUPPORTED)
                refused += 1
        assert refused == 25, refused
        B.jpeg_entropy_decode(data)                          # the whole file: taken
        # the last bytes of the entropy data replaced by EOI: the final MCUs are missing
        with pytest.raises(B.CtpnError):
            B.jpeg_entropy_decode(data[:-12] + b"\xff\xd9")
    assert complete[2]["h"] == 64


def test_too_many_progressive_scans_are_refused():
    """every scan walks every block, so a file of thousands of tiny scans is hours of host work; more than 256 scans are
    CTPN_ERR_UNSUPPORTED. Built from a real progressive file by repeating its first (DC) scan, which delivers the same values again."""
    prog = encode(scene(16, 16, 1), 85, 2, progressive=True)
    offs = _scan_offsets(prog)
    first = prog[offs[0]:offs[1]]                            # the DC scan, marker and data
    a = B.jpeg_entropy_decode(prog)
    b = B.jpeg_entropy_decode(prog[:offs[1]] + first * 200 + prog[offs[1]:])      # redundant, but legal
    assert all(np.array_equal(x, y) for x, y in zip(a[0], b[0]))
    with pytest.raises(B.CtpnError) as e:
        B.jpeg_entropy_decode(prog[:offs[1]] + first * 300 + prog[offs[1]:])
    assert e.value.code == B.CTPN_ERR_UNSUPPORTED and "256" in str(e.value)


def test_a_huffman_table_with_more_codes_than_its_length_holds_is_rejected():
    """Found by fuzzing under ASan: a DHT whose count for a code length exceeds what that many bits can hold (five codes of length 1) made the
    lookahead-table fill of jhuff_build write past the table -- onto the stack, from a crafted file. Such a table is an error now."""
    data = encode(scene(16, 16, 1), 90, 2)
    i = data.index(b"\xff\xc4")
    for length in range(1, 8):                                   # (a count is one byte: from length 8 on it cannot exceed 2^length)
        counts = bytearray(16)
        counts[length - 1] = (1 << length) + 3
        seg = bytes([0x00]) + bytes(counts) + bytes(range(counts[length - 1]))
        bad = data[:i] + b"\xff\xc4" + (len(seg) + 2).to_bytes(2, "big") + seg + data[i:]
        with pytest.raises(B.CtpnError) as e:
            B.jpeg_probe(bad)
        assert e.value.code == -1 and "Huffman" in str(e.value)


def _segments(data):
    """[(marker, body)] of the header part and the rest from the first SOS on (bytes), for syntax-level rewrites of a file."""
    i, segs = 2, []
    while True:
        assert data[i] == 0xFF
        m = data[i + 1]
        if m == 0xDA:
            return segs, data[i:]
        L = int.from_bytes(data[i + 2:i + 4], "big")
        segs.append((m, data[i + 4:i + 2 + L]))
        i += 2 + L


def _assemble(segs, tail, fill=b""):
    out = b"\xff\xd8"
    for m, body in segs:
        out += fill + bytes([0xFF, m]) + (len(body) + 2).to_bytes(2, "big") + body
    return out + fill + tail


@pytest.mark.parametrize("progressive", [False, True], ids=["sequential", "progressive"])
def test_the_same_image_in_other_legal_spellings(progressive):
    """Pillow writes one spelling of the JPEG syntax; the parser has branches it never reaches that way. The same entropy data under rewritten
    headers -- 16-bit quantisation tables (Pq = 1), one DHT segment per table / all tables in one segment, fill bytes (0xFF padding) in front
    of every marker, comment and unknown APPn segments in between, the restart-interval definition moved -- must decode to the same pixels
    (Pillow itself agrees on every rewrite)."""
    for sub, kw in ((2, {}), (0, {"restart_marker_blocks": 3}), (1, {"optimize": True})):
        data = encode(scene(37, 53, 11), 85, sub, progressive=progressive, **kw)
        want = pillow_bgr(data)
        segs, tail = _segments(data)
        rewrites = {}
        # (a) every DQT table in 16-bit form, each in its own segment
        a = []
        for m, body in segs:
            if m != 0xDB:
                a.append((m, body))
                continue
            j = 0
            while j < len(body):
                assert body[j] >> 4 == 0
                a.append((0xDB, bytes([0x10 | (body[j] & 15)]) + b"".join(bytes([0, v]) for v in body[j + 1:j + 65])))
                j += 65
        rewrites["dqt16"] = _assemble(a, tail)
        # (b) one DHT segment per table, and (c) all header tables merged into one segment
        b_, merged = [], b""
        for m, body in segs:
            if m != 0xC4:
                b_.append((m, body))
                continue
            j = 0
            while j < len(body):
                n = sum(body[j + 1:j + 17])
                b_.append((0xC4, body[j:j + 17 + n]))
                merged += body[j:j + 17 + n]
                j += 17 + n
        rewrites["dht-split"] = _assemble(b_, tail)
        if not progressive:          # (a progressive file defines tables between its scans too: merging only the header's ones is still legal)
            c, done = [], False
            for m, body in segs:
                if m == 0xC4:
                    if not done:
                        c.append((0xC4, merged))
                        done = True
                    continue
                c.append((m, body))
            rewrites["dht-merged"] = _assemble(c, tail)
        # (d) fill bytes in front of every header marker; comments and an unknown application segment in between
        rewrites["fill"] = _assemble(segs, tail, fill=b"\xff\xff\xff")
        d = []
        for m, body in segs:
            d += [(m, body), (0xFE, b"a comment"), (0xEB, b"unknown application data \xff\xd8\xff\xda")]
        rewrites["comments"] = _assemble(d, tail)
        # (e) the restart interval defined right behind the frame header instead of where Pillow puts it
        if any(m == 0xDD for m, _ in segs):
            e = [(m, body) for m, body in segs if m != 0xDD]
            k = next(i for i, (m, _) in enumerate(e) if m in (0xC0, 0xC2)) + 1
            rewrites["dri-moved"] = _assemble(e[:k] + [s for s in segs if s[0] == 0xDD] + e[k:], tail)
        # (f) other component identifiers (0, 1, 2 instead of 1, 2, 3) in the frame header and every scan header
        ids = bytearray(data)
        k = max(data.find(b"\xff\xc0"), data.find(b"\xff\xc2"))
        for c in range(3):
            ids[k + 10 + 3 * c] -= 1
        k = 0
        while True:
            k = data.find(b"\xff\xda", k)
            if k < 0:
                break
            for c in range(data[k + 4]):
                ids[k + 5 + 2 * c] -= 1
            k += 4
        rewrites["component-ids"] = bytes(ids)
        for name, blob in rewrites.items():
            assert np.array_equal(pillow_bgr(blob), want), ("Pillow", name)
            planes, qt, lay = B.jpeg_entropy_decode(blob)
            got = J.pixels_from_coefficients(planes, [qt[c] for c in range(lay["ncomp"])], 37, 53, lay["hs"], lay["vs"])
            assert np.array_equal(got, want), (name, sub, progressive)
            assert np.array_equal(J.imread_bgr(blob), want), ("oracle", name)


def test_demo_golden_is_what_its_generator_makes_from_the_reference_tree(golden_dir, tmp_path, monkeypatch):
    """Where the reference tree is at hand (the build container; never on the GPU box), oracle/make_demo_golden.py regenerates
    tests/golden/demo_files.npz array for array: the committed fixture is the reference's own data/demo, not an edited copy."""
    ref = "/root/reference"
    if not os.path.isdir(os.path.join(ref, "data", "demo")):
        pytest.skip("no reference tree here")
    import oracle.make_demo_golden as M
    monkeypatch.setattr(M, "ROOT", str(tmp_path))
    (tmp_path / "tests" / "golden").mkdir(parents=True)
    monkeypatch.setattr("sys.argv", ["make_demo_golden", ref])
    M.main()
    a, b = np.load(os.path.join(golden_dir, "demo_files.npz")), np.load(str(tmp_path / "tests" / "golden" / "demo_files.npz"))
    assert sorted(a.files) == sorted(b.files)
    for k in a.files:
        if k != "decoder":
            assert np.array_equal(a[k], b[k]), k
