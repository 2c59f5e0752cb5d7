"""GPU (-m gpu), round 6 (VERDICT r5 "next" 1 and 8): the parity holes that were still open.

  * BASELINE.json configs[2]'s exact size -- a batch of 32 at 600 x 900 -- equals its images run alone, bit for bit, in the headline's
    arithmetic (bf16) and in the drop-in's default (split); the split batch's images against the fp32 oracle with the gate's assertions;
  * config 5's geometry (1280 x 1920, DETECT_MODE = O) on the bench's own sample seeds (1, 2) as well as the earlier test's (5);
  * the reference's OWN demo images (data/demo/006.jpg .. 010.png, committed bytes) through `ctpn/demo.py` with the shipped config:
    res_<stem>.txt against golden pixels -> oracle/resize_ref.py -> oracle/network.py -> oracle/postproc.py;
  * the two forms of the kernels' LDS-DMA helper (m0 declared clobbered / saved and restored) deliver the same tile bytes.
Nothing here reads /root/reference.
"""
import hashlib
import io
import os

import numpy as np
import pytest

import ctpn_amd
from ctpn_amd import _binding as B
from oracle import network as N
from oracle import postproc as P
from oracle import resize_ref as R
from util import match_lines, match_rois

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def weights(arena):
    return ctpn_amd.arena_views(arena)


def test_lds_dma_helper_forms_agree():
    """conv3x3_impl.h's c3_glds16_saddr puts m0 on its clobber list (two SALU fewer per KiB than saving and restoring it; clang's
    -Winline-asm about reserved registers is silenced for that one statement), c3_glds16_asm saves and restores m0 around the same
    global_load_lds_dwordx4. Both forms on the same tiles: the bytes that arrive in LDS are the source bytes, in both. A compiler that
    starts to keep state in m0 across the statement shows up here (and in every conv layer test) instead of as a silent corruption."""
    rng = np.random.default_rng(6)
    for tiles in (1, 3, 64, 4096 + 5):
        src = rng.integers(0, 256, size=tiles * 1024, dtype=np.uint8)
        a, b = B.debug_lds_dma(src)
        assert np.array_equal(a, src), "m0-clobber form, %d tiles" % tiles
        assert np.array_equal(b, src), "save / restore form, %d tiles" % tiles


@pytest.mark.parametrize("prec", ["bf16", "split"])
def test_batch_of_32_equals_its_images_alone(arena, weights, prec):
    """BASELINE.json configs[2] IS batch 32; the suite's largest end-to-end batch was 8 (16 for the layer tests). The timed configuration
    itself: rois, text lines and heads of a 32-image batch equal those of images 0, 13 and 31 run alone (a batch-1 ctx: other kernels
    for the proposal tail and the ragged columns, the same sums) bit for bit; in split precision two of them are also held to north_star's
    tolerance against the fp32 oracle (the 600 x 900 gate's assertions)."""
    h, w, n = 600, 900, 32
    imgs = ctpn_amd.weights.synthetic_images(n, h, w, 1)           # the bench's images: seeds 1 .. 32
    with ctpn_amd.Context(0, n, h, w, prec) as ctx:
        ctx.load_weights(arena)
        lines, rois = ctx.detect(imgs, want_rois=True)
        cp, bp = ctx.get_tensor("rpn_cls_prob_reshape"), ctx.get_tensor("rpn_bbox_pred")
        # the pipelined form the bench times (two submits in flight) gives the synchronous call's bytes
        ctx.detect_submit(images=imgs, slot=0)
        ctx.detect_submit(images=imgs[::-1].copy(), slot=1)
        l0, r0 = ctx.detect_collect(0, want_rois=True)
        l1, r1 = ctx.detect_collect(1, want_rois=True)
        for i in range(n):
            assert np.array_equal(l0[i], lines[i]) and np.array_equal(r0[i], rois[i]), ("pipelined slot 0", i)
            assert np.array_equal(l1[n - 1 - i], lines[i]) and np.array_equal(r1[n - 1 - i], rois[i]), ("pipelined slot 1", i)
    assert sum(len(l) for l in lines) >= 32
    with ctpn_amd.Context(0, 1, h, w, prec) as ctx:
        ctx.load_weights(arena)
        for i in (0, 13, 31):
            l1, r1 = ctx.detect(imgs[i:i + 1], want_rois=True)
            assert np.array_equal(l1[0], lines[i]) and np.array_equal(r1[0], rois[i]), i
            assert np.array_equal(ctx.get_tensor("rpn_cls_prob_reshape")[0], cp[i]) and np.array_equal(ctx.get_tensor("rpn_bbox_pred")[0], bp[i]), i
    if prec != "split":
        return
    info = np.array([h, w, 1.0], np.float32)
    for i in (13, 31):
        ref = N.forward(imgs[i:i + 1], weights, keep=set())
        d_cls = float(np.abs(cp[i] - ref["rpn_cls_prob_reshape"][0]).max())
        d_box = float(np.abs(bp[i] - ref["rpn_bbox_pred"][0]).max())
        ref_rois = P.proposal_layer(ref["rpn_cls_prob_reshape"], ref["rpn_bbox_pred"], info)
        frac = match_rois(rois[i], ref_rois, px_tol=1.0, score_tol=1e-3)
        ref_lines = P.text_detect(ref_rois[:, 1:5], ref_rois[:, 0], (h, w), "H")
        print("batch 32, image %d, split: cls_prob |diff| %.2e, bbox |diff| %.2e, roi match %.4f, %d lines (oracle %d)"
              % (i, d_cls, d_box, frac, len(lines[i]), len(ref_lines)))
        assert d_cls < 2e-4 and d_box < 1e-3 and frac >= 0.995
        assert match_lines(lines[i], ref_lines, 1.0, 1e-3)


@pytest.mark.parametrize("prec,options", [("bf16", {"nms_prefix": 0, "debug_hog": 5000}), ("bf16", {"nms_prefix": 0, "debug_hog": 7000}),
                                          ("split", {"nms_prefix": 0, "tail_confine": 0}), ("split", {"nms_prefix": 0, "tail_confine": 0, "debug_hog": 3000}),
                                          ("fp16", {"nms_prefix": 0, "debug_hog": 6000}),
                                          # memory-system load (64 workgroups of random 16-byte gathers) beside EVERY layer of the second batch
                                          ("bf16", {"debug_hog": 112000}), ("fp16", {"debug_hog": 112000}), ("split", {"debug_hog": 132000})])
def test_batches_in_flight_do_not_change_each_other(arena, prec, options):
    """The stress that found round 6's race (profiles/r06_barrier_war.txt): two submits in flight, the second one the batch REVERSED, with the
    one-workgroup proposal NMS of the first made long (no prefix pass) and delayed (debug_hog: a spinning kernel of its footprint in front of
    it) so that it runs beside conv3_x / conv4_x / conv5_x of the second -- 1024-thread workgroups that take whole CUs from the persistent conv
    kernels and load the memory system with 16-byte gathers. Before the fix 3 .. 16 of 24 repetitions returned rois of one image changed by up
    to hundreds of pixels (a strip of weights recycled by LDS-DMA under fragment reads still in flight across a barrier); every repetition
    must now give the synchronous call's bytes. Teeth: with the conv kernels put back into their pre-fix form the first three settings fail
    within these 12 repetitions (profiles/r06_barrier_war.txt, 7); the others widen the net (other precisions, pure memory traffic beside
    every layer) and pass there too."""
    h, w, n, reps = 600, 900, 32, 12
    imgs = ctpn_amd.weights.synthetic_images(n, h, w, 1)
    rev = imgs[::-1].copy()
    with ctpn_amd.Context(0, n, h, w, prec, options=options) as ctx:
        ctx.load_weights(arena)
        lines, rois = ctx.detect(imgs, want_rois=True)
        bad = []
        for rep in range(reps):
            ctx.detect_submit(images=imgs, slot=0)
            ctx.detect_submit(images=rev, slot=1)
            l0, r0 = ctx.detect_collect(0, want_rois=True)
            l1, r1 = ctx.detect_collect(1, want_rois=True)
            for i in range(n):
                if not (np.array_equal(r0[i], rois[i]) and np.array_equal(l0[i], lines[i])):
                    bad.append((rep, 0, i))
                if not (np.array_equal(r1[n - 1 - i], rois[i]) and np.array_equal(l1[n - 1 - i], lines[i])):
                    bad.append((rep, 1, i))
    assert not bad, "(repetition, slot, image) whose rois / lines differ from the synchronous call's: %s" % bad[:10]


@pytest.mark.parametrize("prec,n,hog", [("bf16", 1, 101500), ("bf16", 2, 102500), ("split", 1, 103000), ("fp32", 1, 110000), ("fp32", 8, 130000), ("bf16", 8, 104000)])
def test_small_batches_in_flight_under_memory_load(arena, prec, n, hog):
    """The same question for the kernel forms only small batches use (half-tile tails for 8 x 32 patches, 64-pixel flat items, the few-rows
    recurrence, in-stream edge kernels, the multi-workgroup NMS) and for exact fp32: two submits in flight, a gather kernel beside every layer of
    the second, every repetition equal to the synchronous call."""
    h, w, reps = 600, 900, 16
    imgs = ctpn_amd.weights.synthetic_images(n, h, w, 3)
    other = ctpn_amd.weights.synthetic_images(n, h, w, 11)
    with ctpn_amd.Context(0, n, h, w, prec, options={"debug_hog": hog}) as ctx:
        ctx.load_weights(arena)
        lines, rois = ctx.detect(imgs, want_rois=True)
        lo, ro = ctx.detect(other, want_rois=True)
        bad = []
        for rep in range(reps):
            ctx.detect_submit(images=other, slot=0)
            ctx.detect_submit(images=imgs, slot=1)
            l0, r0 = ctx.detect_collect(0, want_rois=True)
            l1, r1 = ctx.detect_collect(1, want_rois=True)
            for i in range(n):
                if not (np.array_equal(r0[i], ro[i]) and np.array_equal(l0[i], lo[i])):
                    bad.append((rep, 0, i))
                if not (np.array_equal(r1[i], rois[i]) and np.array_equal(l1[i], lines[i])):
                    bad.append((rep, 1, i))
    assert not bad, bad[:10]


def test_split_edge_columns_against_the_padded_tile_form(arena):
    """Split precision sends the ragged tile columns of a 600 x 900 image (conv2_x: columns 448, 449 of 450; conv3_1 / conv3_2: column 224 of 225;
    the pooled conv1_2 / conv2_2: the last two pooled columns) through the edge kernel's split form (option split_edge, default 1) instead of a
    padded tile column. Every other output is summed by the same kernel family in the same K order whatever the tile shape: bit-identical between
    the two settings; the edge columns agree to split precision's own accuracy (tap-major against chunk-major K order)."""
    h, w, n = 600, 900, 2
    imgs = ctpn_amd.weights.synthetic_images(n, h, w, 5)
    maps = {}
    for se in (1, 0):
        with ctpn_amd.Context(0, n, h, w, "split", options={"keep_acts": 1, "split_edge": se}) as ctx:
            ctx.load_weights(arena)
            ctx.forward(imgs)
            maps[se] = {nm: ctx.get_tensor(nm) for nm in ("conv1_2", "pool1", "conv2_1", "conv2_2", "pool2", "conv3_1")}
    # first layers: inputs identical in both settings, so interior columns must be bit-identical and edge columns close
    # (conv2_1 reads pool1 through 3 x 3 windows: its own columns from 447 on see pool1's edge columns)
    for nm, edge0, same in (("conv1_2", 896, 896), ("pool1", 448, 448), ("conv2_1", 448, 447)):
        a, b = maps[1][nm], maps[0][nm]
        assert np.array_equal(a[:, :, :same], b[:, :, :same]), nm
        scale = float(np.abs(b).max())
        d = float(np.abs(a[:, :, edge0:] - b[:, :, edge0:]).max())
        print("%s: edge columns %d.. differ by %.2e of max %.2f" % (nm, edge0, d, scale))
        assert a.shape[2] > edge0 and d <= 2e-5 * scale, (nm, d, scale)      # (hi, lo) pairs carry 16 mantissa bits: 2^-17 = 7.6e-6 per rounding
        assert np.abs(a[:, :, edge0:]).max() > 0
    # deeper layers see the tiny edge differences through their 3 x 3 windows: everything within split precision's layer tolerance
    for nm in ("conv2_2", "pool2", "conv3_1"):
        a, b = maps[1][nm], maps[0][nm]
        assert np.abs(a - b).max() <= 1.2e-5 * float(np.abs(b).max()), nm


@pytest.mark.parametrize("seed", [1, 2])
@pytest.mark.parametrize("prec", ["fp32", "split"])
def test_config5_geometry_on_the_bench_sample_seeds(arena, weights, prec, seed):
    """tests/test_gpu_round5.py::test_config5_geometry_end_to_end_against_the_oracle used one image (seed 5); the bench's accuracy sample
    is seeds 1 and 2, and round 5's line reported 127 of 129 split-precision lines within 1 px there (VERDICT r5 "weak" 2). Found
    (tools/r6_config5_knife_edge.py, profiles/r06_config5_knife_edge.json), seed 1, split precision -- two ORDER flips
    between proposals whose scores are one or two fp32 ulps (6e-8) apart in the oracle, where north_star tolerates 1e-3 on scores:
      * column x = 1200: two overlapping proposals (IoU > 0.2) score 0.999340832 and 0.999340713 in the fp32 oracle -- 0.999340832 and
        0.999340773, ONE ulp apart, in a float64 evaluation of the graph. Split precision (cls_prob within 3.4e-5 overall) gives both
        0.999340713: an exact tie, broken by the anchor index, so the connector's NMS 0.2 keeps the other box of the pair; that box
        belongs to another chain, and the two lines involved move by 4.9 and 10.7 px;
      * ranks 999 / 1000 of the 1003 NMS survivors (0.99925977 vs 0.99925965) swap at the post_nms_topN cut: roi 1000 of 1000 differs
        (both candidates are then suppressed by the connector's NMS: no line moves).
    The float64 evaluation itself orders ranks 1000 / 1001 the other way round than the fp32 oracle does. Asserted here, for every seed:
      1. heads within 2e-4 / 1e-3 of the oracle's; the device's rois and lines are EXACTLY the reference-pinned post-processing of the
         device's own heads;
      2. the roi lists differ only by swaps at the top-N cut between scores within 4 ulps (util.topn_cut_swaps refuses anything else);
      3. the SCORES alone explain every line that moved, and they are within 2e-4: the oracle's post-processing on [device cls_prob,
         ORACLE bbox_pred] reproduces the device's lines within 1 px, and on [ORACLE cls_prob, device bbox_pred] the oracle's lines;
      4. every pair of proposals the device orders differently from the oracle is within 4 ulps of a tie in the oracle's scores;
      5. against the plain oracle lines: same count, hull IoU 0.7 for all, at most two lines outside 1 px in split precision, none in fp32
         (order flips inside ties are frequent -- 26 to 88 per image, fp32 device included -- and harmless unless the two boxes overlap)."""
    from util import topn_cut_swaps
    h, w = 1280, 1920
    imgs = ctpn_amd.weights.synthetic_images(1, h, w, seed)
    info1 = np.array([h, w, 1.0], np.float32)
    with ctpn_amd.Context(0, 1, h, w, prec) as ctx:
        ctx.load_weights(arena)
        lines, rois = ctx.detect(imgs, mode="O", want_rois=True, line_capacity=2048)
        cp, bp = ctx.get_tensor("rpn_cls_prob_reshape"), ctx.get_tensor("rpn_bbox_pred")
    ref = N.forward(imgs, weights, keep=set())
    rc, rb = ref["rpn_cls_prob_reshape"], ref["rpn_bbox_pred"]
    d_cls, d_box = float(np.abs(cp[0] - rc[0]).max()), float(np.abs(bp[0] - rb[0]).max())
    assert d_cls < 2e-4 and d_box < 1e-3                                                                   # 1
    exact = P.proposal_layer(cp[0:1], bp[0:1], info1)
    assert rois[0].shape == exact.shape and np.array_equal(rois[0][:, 0], exact[:, 0]) and np.abs(rois[0] - exact).max() < 1e-3
    assert match_lines(lines[0], P.text_detect(exact[:, 1:5], exact[:, 0], (h, w), "O"), 1.0, 1e-3)
    ref_rois = P.proposal_layer(rc, rb, info1)
    ref_ext = P.proposal_layer(rc, rb, info1, post_nms_topn=1016)
    dev_only, ref_only = topn_cut_swaps(rois[0], ref_rois, ref_ext)                                        # 2
    ref_lines = P.text_detect(ref_rois[:, 1:5], ref_rois[:, 0], (h, w), "O")
    hyb = P.proposal_layer(cp[0:1], rb[0:1], info1)                                                        # 3: device scores, oracle boxes
    assert match_lines(lines[0], P.text_detect(hyb[:, 1:5], hyb[:, 0], (h, w), "O"), 1.0, 1e-3)
    hyb = P.proposal_layer(rc[0:1], bp[0:1], info1)                                                        #    oracle scores, device boxes
    assert match_lines(ref_lines, P.text_detect(hyb[:, 1:5], hyb[:, 0], (h, w), "O"), 1.0, 1e-3)
    # 4: position of every oracle roi in the device's list (partner = same x, y within a pixel); inversions against the oracle's order
    pos = np.full(len(ref_rois), -1)
    used = np.zeros(len(rois[0]), bool)
    for i, r in enumerate(ref_rois):
        ok = (np.abs(rois[0][:, 1:5] - r[1:5]).max(axis=1) <= 1.0) & ~used
        if ok.any():
            pos[i] = int(np.argmax(ok)); used[pos[i]] = True
    have = np.where(pos >= 0)[0]
    inv = [(int(i), int(j)) for a, i in enumerate(have) for j in have[a + 1:] if pos[j] < pos[i]]
    ulp = 2.0 ** -24
    for i, j in inv:
        assert abs(float(ref_rois[i, 0]) - float(ref_rois[j, 0])) <= 4 * ulp, (i, j, ref_rois[i], ref_rois[j])
    from bench import _hull_iou_frac, _match_frac                                                          # 5
    lf = _match_frac(lines[0], ref_lines, slice(0, 8), 1.0)
    off = int(round((1.0 - lf) * len(lines[0])))
    print("config 5 geometry, seed %d, %s: cls_prob |diff| %.2e, bbox |diff| %.2e, %d roi swap(s) at the top-N cut, %d order flip(s) within 4 ulps, %d lines (oracle %d), %d outside 1 px"
          % (seed, prec, d_cls, d_box, len(dev_only), len(inv), len(lines[0]), len(ref_lines), off))
    assert len(lines[0]) == len(ref_lines) and _hull_iou_frac(lines[0], ref_lines) == 1.0
    assert off <= (2 if prec == "split" else 0) and (prec == "split" or not dev_only)


def _demo_files(golden_dir):
    g = np.load(os.path.join(golden_dir, "demo_files.npz"))
    return g, [str(nm) for nm in g["names"]]


def _golden_pixels(g, nm):
    """What cv2.imread returns for the committed file: Pillow's decode (libjpeg-turbo / libpng) turned by the EXIF orientation, BGR --
    checked against the SHA-256 oracle/make_demo_golden.py recorded from the reference tree's file."""
    from PIL import Image, ImageOps
    key = nm.replace(".", "_")
    im = Image.open(io.BytesIO(g["file_" + key].tobytes()))
    if im.format == "JPEG":
        im = ImageOps.exif_transpose(im)
    px = np.ascontiguousarray(np.asarray(im.convert("RGB"))[..., ::-1])
    assert hashlib.sha256(px.tobytes()).hexdigest() == str(g["sha256_" + key]), nm
    return px


def test_reference_demo_images_end_to_end_against_the_oracle(tmp_path, arena, weights, golden_dir):
    """`python ctpn/demo.py` (reference ctpn/demo.py:55-68,99-104) on the reference's OWN data/demo files, shipped config (split
    precision, DETECT_MODE H): imread -> resize_im (factors 0.625, 0.8798, 1.0 on the EXIF-turned 008, 0.625, 2.4: four non-identity
    resizes, one of them to an odd width, 901) -> second rescale (identity for all five) -> network -> proposal layer -> connector ->
    res_<stem>.txt. Against: golden pixels -> oracle/resize_ref.py -> oracle/network.py (fp32) -> oracle/postproc.py, whose
    draw_boxes text must equal the written file BYTE FOR BYTE. Natural images, not noise: large flat regions, saturated pixels."""
    pytest.importorskip("PIL")
    from ctpn_amd.ctpn import demo
    from ctpn_amd.lib.fast_rcnn.config import cfg
    g, names = _demo_files(golden_dir)
    root = tmp_path
    (root / "data" / "demo").mkdir(parents=True)
    for nm in names:
        (root / "data" / "demo" / nm).write_bytes(g["file_" + nm.replace(".", "_")].tobytes())
    cwd = os.getcwd()
    try:
        demo.main(["--root", str(root), "--synthetic", "0"])
        assert cfg.TEST.PRECISION == "split" and cfg.TEST.DETECT_MODE == "H"
    finally:
        os.chdir(cwd)
    n_lines, report = 0, []
    for nm in names:
        px = _golden_pixels(g, nm)
        f = demo.resize_factor(px.shape, 600, 1200)
        im = R.resize_linear(px, f, f) if f != 1.0 else px
        h, w = im.shape[:2]
        assert (h, w) == B.resize_dims(px.shape[0], px.shape[1], f, f)
        assert min(h, w) == 600 and max(h, w) <= 1000          # the second rescale (lib/fast_rcnn/test.py:17-24) is the identity for these files
        ref = N.forward(im[None], weights, keep=set())
        info = np.array([h, w, 1.0], np.float32)
        rois = P.proposal_layer(ref["rpn_cls_prob_reshape"], ref["rpn_bbox_pred"], info)
        recs = P.text_detect(rois[:, 1:5], rois[:, 0], (h, w), "H")
        want = "".join(P.draw_boxes_lines(recs, f)).encode()
        stem = nm.split(".")[0]
        got = (root / "data" / "results" / ("res_%s.txt" % stem)).read_bytes()
        report.append("%s %dx%d f=%.4f -> %dx%d: %d rois, %d lines, file %s" % (nm, px.shape[0], px.shape[1], f, h, w, len(rois), got.count(b"\r\n"),
                                                                              "EQUAL" if got == want else "DIFFERS"))
        n_lines += got.count(b"\r\n")
        assert got == want, "%s: res_%s.txt differs from the oracle path's\n got  %r\n want %r" % (nm, stem, got[:400], want[:400])
        # the annotated image is written at the ORIGINAL size (draw_boxes resizes back by 1 / f: demo.py:51-52), like the reference's own
        from PIL import Image
        with Image.open(str(root / "data" / "results" / nm)) as out:
            assert (out.size[1], out.size[0]) == tuple(int(v) for v in g["result_hw_" + nm.replace(".", "_")]), nm
    print("\n".join(report))
    assert n_lines >= 5        # the comparison is not vacuous


def test_nms_prefix_pass_equals_the_full_pass():
    """Option nms_prefix (round 6, default on): the proposal layer's column NMS first looks at the 4096 best-scored candidates; they hold the
    post_nms_topN survivors unless few survive, in which case a full pass follows (in-kernel for the one-workgroup-per-image form, a second
    launch for the multi-workgroup form). Whether rank r survives depends on ranks below r only, so the keep list is the same by construction;
    here, for both forms and both outcomes of the prefix pass: rois and anchors with nms_prefix = 1 == nms_prefix = 0 == the generic NMS ==
    the oracle's proposal_layer, on (a) ordinary heads (the prefix answers), (b) "tall" heads (dh = 3: every box of a column is clipped to
    the full image height, ONE survivor per column, 56 in all: the prefix cannot answer, the full pass must run), (c) half the columns
    tall, (d) top-N values around the prefix length."""
    rng = np.random.default_rng(66)
    hf, wf = 37, 56
    cases = [("ordinary", 0.7, 12000, 1000), ("tall", 0.7, 12000, 1000), ("half-tall", 0.7, 12000, 1000), ("ordinary-0.3", 0.3, 12000, 1000),
             ("small-topn", 0.7, 12000, 300), ("pre-6000", 0.7, 6000, 1000), ("pre-4096", 0.7, 4096, 1000), ("tall-pre-5000", 0.7, 5000, 1000)]
    fallbacks = 0
    for name, thr, pre, post in cases:
        for n in (1, 4, 8):
            fg = rng.random((n, hf, wf, 10), dtype=np.float32)
            cls = np.zeros((n, hf, wf, 20), np.float32)
            cls[..., 1::2] = fg
            cls[..., 0::2] = 1.0 - fg
            bbox = (rng.standard_normal((n, hf, wf, 40)) * 0.4).astype(np.float32)
            if name.startswith("tall"):
                bbox[..., 3::4] = 3.0
            elif name == "half-tall":
                bbox[:, :, 0::2, 3::4] = 3.0
            info = np.array([[hf * 16, wf * 16, 1.0]] * n, np.float32)
            got = {}
            for form, prefix in ((1, 1), (1, 0), (2, 1), (2, 0), (3, 1), (0, 0)):
                with ctpn_amd.Context(0, 8, hf * 16, wf * 16, "fp32", postproc_only=True, options={"nms_columns": form, "nms_prefix": prefix}) as ctx:
                    got[(form, prefix)] = ctx.proposals_from_host(cls, bbox, info, pre, post, thr, 8.0, want_anchors=True)
            base = got[(0, 0)]
            for key, val in got.items():
                for i in range(n):
                    assert val[0][i].shape == base[0][i].shape and np.array_equal(val[0][i], base[0][i]), (name, n, key, i, "rois")
                    assert np.array_equal(val[1][i], base[1][i]), (name, n, key, i, "anchors")
            want = P.proposal_layer(cls[:1], bbox[:1], info[0], pre_nms_topn=pre, post_nms_topn=post, nms_thresh=thr, min_size=8.0)
            r0 = base[0][0]
            assert r0.shape == want.shape and (r0.size == 0 or np.abs(r0 - want).max() < 1e-3), (name, n)
            fallbacks += int(len(r0) < post)          # fewer rois than asked for: the prefix pass could not have answered
    assert fallbacks >= 6
