"""CPU tests of the JPEG decoder's seams (cv2.imread in front of the hot path, reference ctpn/demo.py:59).

  Pillow (libjpeg-turbo; the pin)  ==  oracle/jpeg_ref.py, whole pipeline in numpy                              (oracle pinned)
  library host half (ctpn_jpeg_entropy_decode, C++)  ==  oracle entropy decoder, coefficient for coefficient       (bit-exact)
  library host half -> oracle pixel half  ==  Pillow                                                             (bit-exact)

The device half (IDCT / upsampling / colour kernels) is the GPU suite's: tests/test_gpu_jpeg.py. Nothing here needs a GPU or
/root/reference.
"""
import io

import numpy as np
import pytest

import ctpn_amd  # noqa: F401
from ctpn_amd import _binding as B
from oracle import jpeg_ref as J
from util_jpeg import CASES, case_id, encode, pillow_bgr, scene


def test_committed_vectors(golden_dir):
    """tests/golden/jpeg_cases.npz (oracle/make_jpeg_golden.py: files + libjpeg-turbo's decode of them): the oracle, the library's host
    half in front of the oracle's pixel half, and the Pillow installed here all reproduce the committed pixels."""
    import os
    g = np.load(os.path.join(golden_dir, "jpeg_cases.npz"))
    assert len(g["names"]) == len(CASES)
    for name in g["names"]:
        data, want = g["file_" + name].tobytes(), g["bgr_" + name]
        assert np.array_equal(J.imread_bgr(data), want), name
        planes, qt, lay = B.jpeg_entropy_decode(data)
        got = J.pixels_from_coefficients(planes, [qt[c] for c in range(lay["ncomp"])], lay["h"], lay["w"], lay["hs"])
        assert np.array_equal(got, want), name
        assert np.array_equal(pillow_bgr(data), want), name


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
    assert lay["hs"] == (1 if gray or sub == 0 else 2)
    assert len(planes) == len(want)
    for c, (g, o) in enumerate(zip(planes, want)):
        assert g.shape == o.shape, (c, g.shape, o.shape)
        assert np.array_equal(g.astype(np.int32), o), c
    for c, comp in enumerate(frame["comps"]):
        assert np.array_equal(qt[c].astype(np.int64), frame["qt"][comp[3]])


# the sizes the hot path is quoted on (BASELINE.json configs[1]) and its neighbours: the library's host half in front of the vectorised oracle
@pytest.mark.parametrize("geom", [(600, 900, 90, 2), (600, 900, 75, 0), (601, 899, 95, 2), (255, 1201, 60, 2)], ids=lambda g: "%dx%d-q%d-s%d" % g)
def test_host_half_then_oracle_pixels_equals_pillow_at_full_size(geom):
    h, w, q, sub = geom
    data = encode(scene(h, w, 5), q, sub)
    planes, qt, lay = B.jpeg_entropy_decode(data)
    got = J.pixels_from_coefficients(planes, [qt[c] for c in range(lay["ncomp"])], h, w, lay["hs"])
    want = pillow_bgr(data)
    assert np.array_equal(got, want)


def test_random_sizes_qualities_and_layouts_equal_pillow():
    """Eighty files of random size (every partial-MCU case, the narrow images whose chroma libjpeg replicates), quality 1..100 (quantisation
    tables from all-255 to all-1), layout, optimised tables, restart intervals."""
    rng = np.random.default_rng(7)
    for k in range(80):
        h, w = int(rng.integers(1, 120)), int(rng.integers(1, 120))
        q, sub, gray = int(rng.integers(1, 101)), int(rng.choice([0, 2])), bool(rng.integers(0, 5) == 0)
        kw = {"optimize": True} if k % 3 == 0 else ({"restart_marker_blocks": int(rng.integers(1, 9))} if k % 3 == 1 else {})
        data = encode(scene(h, w, k, gray), q, sub, **kw)
        planes, qt, lay = B.jpeg_entropy_decode(data)
        got = J.pixels_from_coefficients(planes, [qt[c] for c in range(lay["ncomp"])], h, w, lay["hs"])
        assert np.array_equal(got, pillow_bgr(data)), (k, h, w, q, sub, gray, kw)


def test_probe_reads_the_header_only():
    data = encode(scene(37, 53, 1), 90, 2)
    assert B.jpeg_probe(data) == (37, 53, 3, 2)
    assert B.jpeg_probe(data[: data.index(b"\xff\xda") + 14])[:2] == (37, 53)      # everything up to the scan header is enough
    assert B.jpeg_probe(encode(scene(20, 30, 1, gray=True))) == (20, 30, 1, 1)
    assert B.jpeg_probe(encode(scene(20, 30, 1), 90, 0)) == (20, 30, 3, 1)


def test_probe_files_scans_a_directory_in_one_call(tmp_path):
    """ctpn_jpeg_probe_files: per-file outcomes are data (h = 0: not for the device decoder), never an error of the call."""
    from PIL import Image
    good = encode(scene(37, 53, 1), 90, 2)
    files = {
        "a.jpg": good,
        "b.jpg": encode(scene(20, 30, 2), 80, 0),
        "c.jpg": encode(scene(24, 40, 3, gray=True), 80),
        "d.jpg": encode(scene(40, 56, 4), 90, 2, progressive=True),                      # unsupported kind
        "e.jpg": b"not a jpeg at all",
        # 150 KB of APP1 segments in front of the frame header: more than the 64 KB the scan reads first
        "f.jpg": good[:2] + b"".join(b"\xff\xe1" + (50002).to_bytes(2, "big") + bytes(50000) for _ in range(3)) + good[2:],
    }
    for k, v in files.items():
        (tmp_path / k).write_bytes(v)
    Image.fromarray(scene(16, 16, 5)).save(str(tmp_path / "g.png"))
    names = [str(tmp_path / k) for k in ("a.jpg", "b.jpg", "c.jpg", "d.jpg", "e.jpg", "f.jpg", "g.png", "missing.jpg")]
    for threads in (0, 1, 3):
        got = B.jpeg_probe_files(names, threads)
        assert got.tolist() == [[37, 53, 3, 2], [20, 30, 3, 1], [24, 40, 1, 1], [0, 0, 0, 0], [0, 0, 0, 0], [37, 53, 3, 2], [0, 0, 0, 0], [0, 0, 0, 0]]
    assert B.jpeg_probe_files([]).shape == (0, 4)
    # and the file with the long header decodes like the plain one (Pillow agrees)
    planes, qt, lay = B.jpeg_entropy_decode(files["f.jpg"])
    assert np.array_equal(J.pixels_from_coefficients(planes, [qt[c] for c in range(3)], 37, 53, 2), pillow_bgr(files["f.jpg"]))


@pytest.mark.parametrize("kw", [{"progressive": True}, {"subsampling": 1}], ids=["progressive", "422"])
def test_files_of_other_kinds_are_reported_as_unsupported_not_decoded_wrongly(kw):
    sub = kw.pop("subsampling", 2)
    data = encode(scene(40, 56, 2), 90, sub, **kw)
    with pytest.raises(B.CtpnError) as e:
        B.jpeg_probe(data)
    assert e.value.code == B.CTPN_ERR_UNSUPPORTED
    with pytest.raises(B.CtpnError) as e:
        B.jpeg_entropy_decode(data)
    assert e.value.code == B.CTPN_ERR_UNSUPPORTED


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
    # cut inside the entropy-coded segment: libjpeg pads with zero bits and warns; the library decodes what is there (no crash, no read
    # past the end -- the ASan build runs this too) and whatever it returns has the right shape
    planes, qt, lay = B.jpeg_entropy_decode(data[: len(data) // 2])
    assert planes[0].shape == (8, 8, 64)
    # random bytes after a valid header: either an error or garbage coefficients, never a crash
    rng = np.random.default_rng(0)
    head = data[: data.index(b"\xff\xda") + 14]
    for _ in range(20):
        junk = head + rng.integers(0, 256, 600, dtype=np.uint8).tobytes()
        try:
            B.jpeg_entropy_decode(junk)
        except B.CtpnError as e:
            assert e.code in (-1, B.CTPN_ERR_UNSUPPORTED)


def test_coefficient_capacity_covers_every_supported_layout():
    lib = B.load_library()
    for (h, w) in [(1, 1), (8, 8), (9, 9), (600, 900), (601, 899), (17, 1201)]:
        cap = lib.ctpn_jpeg_coef_capacity(h, w)
        for sub, gray in [(0, False), (2, False), (2, True)]:
            planes, _, _ = B.jpeg_entropy_decode(encode(scene(h, w, 1, gray), 50, sub))
            assert sum(p.size for p in planes) <= cap
    assert lib.ctpn_jpeg_coef_capacity(0, 10) == 0
