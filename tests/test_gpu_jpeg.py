"""GPU (-m gpu): cv2.imread + resize_im on the device (reference ctpn/demo.py:59-60) through the C ABI.

ctpn_decode_jpeg_batch = host entropy decoding (checked on the CPU against the oracle: tests/test_jpeg.py) + jpeg_idct_kernel +
jpeg_color_kernel (+ the resize kernel). The bar is byte equality with Pillow's decode of the same file (libjpeg-turbo: the decoder family
behind cv2.imread, see oracle/jpeg_ref.py), for every layout the decoder takes, odd sizes included; and the decoded batch, handed to the
detector as a device pointer, must give the lines the host-decoded pixels give. Nothing here reads /root/reference.
"""
import os

import numpy as np
import pytest

import ctpn_amd
from ctpn_amd import _binding as B
from util_jpeg import CASES, case_id, cv2_like_bgr, encode, encode_custom, pillow_bgr, scene, with_exif_orientation, with_luma_sampling

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(arena):
    with ctpn_amd.Context(0, 8, 600, 900, "bf16") as c:
        c.load_weights(arena)
        yield c


@pytest.mark.parametrize("case", CASES, ids=case_id)
def test_device_decode_equals_pillow(ctx, case):
    h, w, q, sub, gray, kw = case
    data = encode(scene(h, w, h + w, gray), q, sub, **kw)
    ptr, shape = ctx.decode_jpeg_batch([data])
    assert shape == (1, h, w)
    got = ctx.jpeg_batch_fetch(ptr, shape)[0]
    want = pillow_bgr(data)
    d = np.argwhere(got != want)
    assert d.size == 0, "%d bytes differ, first at %s: %d vs %d" % (len(d), d[0], got[tuple(d[0])], want[tuple(d[0])])


def test_device_decode_equals_the_committed_vectors(ctx, golden_dir):
    """tests/golden/jpeg_cases.npz: files and libjpeg-turbo's pixels as committed (independent of the Pillow installed on this box)."""
    g = np.load(os.path.join(golden_dir, "jpeg_cases.npz"))
    for name in g["names"]:
        data, want = g["file_" + name].tobytes(), g["bgr_" + name]
        ptr, shape = ctx.decode_jpeg_batch([data])
        assert np.array_equal(ctx.jpeg_batch_fetch(ptr, shape)[0], want), name


def test_random_geometries_and_qualities_equal_pillow(ctx):
    """Sixty files of random size (1..200 in both directions: every partial-MCU case), quality and layout (4:4:4 / 4:2:2 / 4:2:0 / gray),
    every fourth one progressive."""
    rng = np.random.default_rng(2024)
    for k in range(60):
        h, w = int(rng.integers(1, 200)), int(rng.integers(1, 200))
        q, sub, gray = int(rng.integers(5, 101)), int(rng.choice([0, 1, 2])), bool(rng.integers(0, 5) == 0)
        kw = {"optimize": True} if k % 3 == 0 else ({"restart_marker_blocks": int(rng.integers(1, 9))} if k % 3 == 1 else {})
        if k % 4 == 3:
            kw["progressive"] = True
        data = encode(scene(h, w, k, gray), q, sub, **kw)
        ptr, shape = ctx.decode_jpeg_batch([data])
        got = ctx.jpeg_batch_fetch(ptr, shape)[0]
        assert np.array_equal(got, pillow_bgr(data)), (k, h, w, q, sub, gray, kw)


@pytest.mark.parametrize("sub", [2, 0, 1], ids=["420", "444", "422"])
def test_a_batch_at_the_benchmark_geometry_equals_pillow(ctx, sub):
    """... sequential and progressive files side by side in one batch: they differ on the host half only."""
    datas = [encode(scene(600, 900, 100 + i), 90, sub, progressive=bool(i & 1)) for i in range(8)]
    ptr, shape = ctx.decode_jpeg_batch(datas, 600, 900)
    assert shape == (8, 600, 900)
    got = ctx.jpeg_batch_fetch(ptr, shape)
    for i, d in enumerate(datas):
        assert np.array_equal(got[i], pillow_bgr(d)), i


@pytest.mark.parametrize("geom", [(300, 450, 2.0), (900, 1350, 600.0 / 900.0), (123, 457, 600.0 / 123.0 if 457 * 600.0 / 123.0 <= 1200 else 1200.0 / 457.0)],
                         ids=["up2", "down-from-larger-than-the-ctx", "odd"])
def test_decode_with_resize_equals_resize_of_the_decoded_image(ctx, geom):
    """resize_im in the same call: the file may be larger than the ctx's network capacity (its buffers follow the file size)."""
    h, w, f = geom
    datas = [encode(scene(h, w, 7 + i), 88, 2) for i in range(2)]
    ptr, shape = ctx.decode_jpeg_batch(datas, h, w, f, f)
    want = np.stack([B.resize_linear(pillow_bgr(d), f, f) for d in datas])
    assert shape == want.shape[:3]
    assert np.array_equal(ctx.jpeg_batch_fetch(ptr, shape), want)


def test_decoded_batches_feed_the_detector_as_the_host_pixels_do(ctx):
    """Three batches in flight the way ctpn/demo_batch.py --decode gpu drives them (decode k + 1 is queued while batch k's forward runs,
    the third decode reuses the first one's buffers): same lines as detect() on Pillow's pixels, and the buffers' reuse corrupts nothing."""
    batches = [[encode(scene(256, 384, 10 * b + i), 90, 2) for i in range(4)] for b in range(3)]
    want = [ctx.detect(np.stack([pillow_bgr(d) for d in datas]), mode="H") for datas in batches]
    got, pending = [], None
    for k, datas in enumerate(batches):
        ptr, shape = ctx.decode_jpeg_batch(datas, 256, 384)
        ctx.detect_submit(device_ptr=ptr, shape=shape, slot=k & 1)
        if pending is not None:
            got.append(ctx.detect_collect(pending, mode="H"))
        pending = k & 1
    got.append(ctx.detect_collect(pending, mode="H"))
    assert sum(len(x) for b in want for x in b) > 0
    for b in range(3):
        for i in range(4):
            assert np.array_equal(got[b][i], want[b][i]), (b, i)


def test_decode_from_paths_equals_decode_from_memory(ctx, tmp_path):
    datas = [encode(scene(120, 200, 60 + i), 85, 2) for i in range(5)]
    names = []
    for i, d in enumerate(datas):
        (tmp_path / ("p%d.jpg" % i)).write_bytes(d)
        names.append(str(tmp_path / ("p%d.jpg" % i)))
    ptr, shape = ctx.decode_jpeg_files(names, 120, 200)
    got = ctx.jpeg_batch_fetch(ptr, shape)
    for i, d in enumerate(datas):
        assert np.array_equal(got[i], pillow_bgr(d)), i
    with pytest.raises(B.CtpnError) as e:
        ctx.decode_jpeg_files(names + [str(tmp_path / "missing.jpg")], 120, 200)
    assert e.value.code == -1 and "missing.jpg" in str(e.value)


def test_buffers_grow_and_live_batches_survive(ctx):
    small = [encode(scene(40, 56, 1), 90, 2)]
    big = [encode(scene(700, 1100, 2 + i), 80, 0) for i in range(3)]
    p1, s1 = ctx.decode_jpeg_batch(small)
    p2, s2 = ctx.decode_jpeg_batch(big)                    # the other buffer set, grown
    assert np.array_equal(ctx.jpeg_batch_fetch(p1, s1)[0], pillow_bgr(small[0]))
    got = ctx.jpeg_batch_fetch(p2, s2)
    for i, d in enumerate(big):
        assert np.array_equal(got[i], pillow_bgr(d))
    p3, s3 = ctx.decode_jpeg_batch(big)                    # the first set again, grown in turn
    assert np.array_equal(ctx.jpeg_batch_fetch(p3, s3)[2], pillow_bgr(big[2]))
    with pytest.raises(B.CtpnError):
        ctx.jpeg_batch_fetch(p3 + 64, s3)


def test_argument_and_layout_errors(ctx):
    a, b = encode(scene(48, 64, 1), 90, 2), encode(scene(48, 64, 2), 90, 0)
    with pytest.raises(B.CtpnError) as e:
        ctx.decode_jpeg_batch([a, b], 48, 64)              # 4:2:0 and 4:4:4 in one batch
    assert e.value.code == B.CTPN_ERR_UNSUPPORTED
    with pytest.raises(B.CtpnError) as e:
        ctx.decode_jpeg_batch([a], 48, 72)                 # not the announced size
    assert e.value.code == -1
    with pytest.raises(B.CtpnError) as e:
        ctx.decode_jpeg_batch([with_luma_sampling(encode(scene(48, 64, 1), 90, 2), 0x41)], 48, 64)      # 4:1:1: not a layout the decoder takes
    assert e.value.code == B.CTPN_ERR_UNSUPPORTED
    with pytest.raises(B.CtpnError) as e:
        ctx.decode_jpeg_batch([a, encode(scene(48, 64, 3), 90, 1)], 48, 64)                  # 4:2:0 and 4:2:2 in one batch
    assert e.value.code == B.CTPN_ERR_UNSUPPORTED
    with pytest.raises(B.CtpnError) as e:
        ctx.decode_jpeg_batch([with_exif_orientation(a, 3), a], 48, 64)                      # two EXIF orientations in one batch
    assert e.value.code == B.CTPN_ERR_UNSUPPORTED
    with pytest.raises(B.CtpnError) as e:
        ctx.decode_jpeg_batch([a[: len(a) // 2]], 48, 64)                                    # truncated: libjpeg's rules, the host decoder's file
    assert e.value.code == B.CTPN_ERR_UNSUPPORTED
    ptr, shape = ctx.decode_jpeg_batch([a], 48, 64)        # and the ctx is still usable
    assert np.array_equal(ctx.jpeg_batch_fetch(ptr, shape)[0], pillow_bgr(a))


def test_440_files_and_exif_orientations_on_the_device(ctx):
    """Round 5: 4:4:0 (jdsample.c's h1v2 filter in jpeg_pixel) and the eight EXIF orientations (the colour kernel's index
    map) against what cv2.imread returns = Pillow's decode turned by ImageOps.exif_transpose. Random sizes, every layout, sequential and
    progressive, batches of several files; the probe reports the turned size."""
    rng = np.random.default_rng(5)
    for k in range(24):
        h, w = int(rng.integers(1, 150)), int(rng.integers(1, 150))
        data = encode_custom(scene(h, w, k), 1, 2, q=int(rng.integers(2, 30)), restart=int(rng.integers(0, 4)))
        ptr, shape = ctx.decode_jpeg_batch([data])
        assert shape == (1, h, w) and np.array_equal(ctx.jpeg_batch_fetch(ptr, shape)[0], pillow_bgr(data)), ("440", k, h, w)
    for k in range(48):
        h, w, o = int(rng.integers(1, 120)), int(rng.integers(1, 120)), 1 + k % 8
        sub = (k // 8) % 4
        data = encode_custom(scene(h, w, k), 1, 2, orientation=o) if sub == 3 else with_exif_orientation(encode(scene(h, w, k), 85, sub, progressive=bool(k & 16)), o, bool(k & 32))
        want = cv2_like_bgr(data)
        assert B.jpeg_probe(data)[:2] == want.shape[:2]
        ptr, shape = ctx.decode_jpeg_batch([data])
        assert shape == (1,) + want.shape[:2] and np.array_equal(ctx.jpeg_batch_fetch(ptr, shape)[0], want), ("orientation", o, sub, h, w)
    # a batch of turned files at the benchmark geometry (stored 900 x 600, shown 600 x 900), decode + resize_im in one call
    datas = [with_exif_orientation(encode(scene(900, 600, 70 + i), 90, 2), 6) for i in range(4)]
    ptr, shape = ctx.decode_jpeg_batch(datas, 600, 900)
    got = ctx.jpeg_batch_fetch(ptr, shape)
    for i, d in enumerate(datas):
        assert np.array_equal(got[i], cv2_like_bgr(d)), i
    ptr, shape = ctx.decode_jpeg_batch(datas[:2], 600, 900, 0.5, 0.5)
    assert shape == (2, 300, 450)
    want = B.resize_linear(np.stack([cv2_like_bgr(d) for d in datas[:2]]), 0.5, 0.5)
    assert np.array_equal(ctx.jpeg_batch_fetch(ptr, shape), want)


def test_the_references_own_demo_files_go_through_the_device_decoder(tmp_path, arena, golden_dir):
    """data/demo/006.jpg .. 009.jpg + 010.png of the reference tree (ctpn/demo.py:59 reads them; committed with the SHA-256 of what
    cv2.imread returns for each by oracle/make_demo_golden.py): 006 and 009 are 4:4:0, 008 carries EXIF orientation 6 -- round 4 sent all
    three to Pillow. Now: the header scan takes all four JPEG files, the device decode equals the committed pixels, the PNG decodes through
    the library's host decoder, and `demo_batch --decode gpu` over a copy of the directory routes NOTHING to Pillow and writes the result
    files the host-decode path writes."""
    import hashlib
    from ctpn_amd.ctpn import demo_batch
    from ctpn_amd.lib.fast_rcnn.config import cfg
    from ctpn_amd.lib.networks.factory import get_network
    g = np.load(os.path.join(golden_dir, "demo_files.npz"))
    src = tmp_path / "demo"
    src.mkdir()
    for nm in g["names"]:
        (src / str(nm)).write_bytes(g["file_" + str(nm).replace(".", "_")].tobytes())
    jpgs = [str(src / str(nm)) for nm in g["names"] if str(nm).endswith(".jpg")]
    probed = B.jpeg_probe_files(jpgs)
    assert (probed[:, 0] > 0).all(), probed
    with ctpn_amd.Context(0, 1, 600, 900, "bf16") as c:
        c.load_weights(arena)
        for path, pr in zip(jpgs, probed.tolist()):
            key = os.path.basename(path).replace(".", "_")
            assert tuple(pr[:2]) == tuple(g["shape_" + key][:2])
            ptr, shape = c.decode_jpeg_files([path], pr[0], pr[1])
            got = c.jpeg_batch_fetch(ptr, shape)[0]
            assert np.array_equal(got[:32, :32], g["windows_" + key][0]) and np.array_equal(got[-32:, -32:], g["windows_" + key][3]), key
            assert hashlib.sha256(got.tobytes()).hexdigest() == str(g["sha256_" + key]), key
    png = g["file_010_png"].tobytes()
    assert hashlib.sha256(B.png_decode(png).tobytes()).hexdigest() == str(g["sha256_010_png"])
    cfg.TEST.PRECISION = "bf16"
    net = get_network("VGGnet_test")
    net.load_arena(arena)
    try:
        names = demo_batch.list_images(str(src))
        assert len(names) == 5
        logs = []
        res_g = demo_batch.run(net, names, str(tmp_path / "gpu"), batch=4, write_images=True, log=logs.append, decode="gpu")
        res_h = demo_batch.run(net, names, str(tmp_path / "host"), batch=4, write_images=True, log=lambda *_: None)
        assert "4 decoded on the device, 1 PNG files by the library, 0 on the host" in logs[0], logs
        for nm in names:
            assert np.array_equal(res_g[nm], res_h[nm]), nm
            stem = os.path.basename(nm).split(".")[0]
            assert (tmp_path / "gpu" / ("res_%s.txt" % stem)).read_bytes() == (tmp_path / "host" / ("res_%s.txt" % stem)).read_bytes(), stem
    finally:
        net.close()


def test_batch_cli_with_device_decode_writes_the_host_decode_paths_files(tmp_path, arena):
    """ctpn/demo_batch.py --decode gpu against its default (Pillow) path on a directory of mixed sizes and kinds: JPEG 4:2:0 / 4:4:4 at
    sizes that need resize_im both ways, a progressive file and a 4:2:2 file (device decoder), two PNG files (the library's host decoder:
    one at the network's size, an RGBA one that resize_im enlarges), a CMYK JPEG and a 16-bit PNG (Pillow): identical res_<stem>.txt,
    identical annotated images."""
    from PIL import Image
    from ctpn_amd.ctpn import demo_batch
    from ctpn_amd.lib.fast_rcnn.config import cfg
    from ctpn_amd.lib.networks.factory import get_network
    src, out_g, out_h = tmp_path / "in", tmp_path / "gpu", tmp_path / "host"
    src.mkdir()
    files = [(300, 450, 2, {}), (300, 450, 2, {}), (600, 900, 2, {}), (300, 450, 0, {}), (700, 1050, 2, {}), (300, 450, 2, {"progressive": True}),
             (300, 450, 1, {})]
    for i, (h, w, sub, kw) in enumerate(files):
        (src / ("im%02d.jpg" % i)).write_bytes(encode(scene(h, w, 40 + i), 90, sub, **kw))
    Image.fromarray(scene(300, 450, 99)).save(str(src / "im99.png"))
    Image.fromarray(scene(300, 450, 98)).convert("CMYK").save(str(src / "im98.jpg"), "JPEG", quality=90)
    Image.fromarray(np.dstack([scene(200, 300, 97), scene(200, 300, 96, gray=True)])).save(str(src / "im97.png"))
    Image.fromarray(scene(300, 450, 95, gray=True).astype(np.uint16) * 257).save(str(src / "im95.png"))
    cfg.TEST.PRECISION = "bf16"
    net = get_network("VGGnet_test")
    net.load_arena(arena)
    try:
        names = demo_batch.list_images(str(src))
        logs = []
        res_g = demo_batch.run(net, names, str(out_g), batch=4, write_images=True, log=logs.append, decode="gpu")
        res_h = demo_batch.run(net, names, str(out_h), batch=4, write_images=True, log=lambda *_: None)
        assert "7 decoded on the device, 2 PNG files by the library, 2 on the host" in logs[0], logs
        for nm in names:
            assert np.array_equal(res_g[nm], res_h[nm]), nm
            base = os.path.basename(nm)
            stem = base.split(".")[0]
            assert (out_g / ("res_%s.txt" % stem)).read_bytes() == (out_h / ("res_%s.txt" % stem)).read_bytes(), stem
            assert np.array_equal(np.asarray(Image.open(str(out_g / base))), np.asarray(Image.open(str(out_h / base)))), base
    finally:
        net.close()
