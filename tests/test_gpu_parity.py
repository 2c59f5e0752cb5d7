"""GPU (-m gpu): the HIP path through the C ABI against the oracle and the reference-generated fixtures.

Bars (BASELINE.json north_star): scores within 1e-3 (fp32 path), box coordinates +-1 px, NMS keep lists bit-exact
for identical sorted inputs, tie order documented (descending score, ties by ascending index). bf16 conv path:
reported against the same oracle with the looser tolerance written in each test.
Nothing here reads /root/reference.
"""
import os

import numpy as np
import pytest

import ctpn_amd
from ctpn_amd import _binding as B
from oracle import network as N
from oracle import postproc as P
from oracle.make_golden import CASES, synth_inputs
from util import TIE_HEAVY, canon_rows, lines_close, match_lines, match_rois

pytestmark = pytest.mark.gpu

SMALL = (2, 70, 100)   # odd sizes: M tails in every GEMM, VALID pools dropping rows/cols


@pytest.fixture(scope="module")
def weights(arena):
    return ctpn_amd.arena_views(arena)


def rel_err(got, ref):
    return float(np.abs(np.asarray(got, np.float64) - ref).max() / max(float(np.abs(ref).max()), 1e-30))


def test_library_is_the_hip_one_and_single_runtime():
    assert B.device_count() >= 1
    maps = open("/proc/self/maps").read()
    assert "libctpn_hip.so" in maps
    hips = {l.split()[-1] for l in maps.splitlines() if "libamdhip64" in l}
    assert len(hips) == 1, "two HIP runtimes loaded: %s" % hips


@pytest.fixture(autouse=True)
def _default_kernel_selection():
    # (the binding maps CTPN_KEEP_ACTS & co. onto ctpn_set_option of every Context it creates; the library reads no such variable)
    os.environ["CTPN_KEEP_ACTS"] = "0"
    yield
    os.environ["CTPN_KEEP_ACTS"] = "0"


def test_fp32_every_layer_matches_oracle(arena, weights):
    os.environ["CTPN_KEEP_ACTS"] = "1"
    n, h, w = SMALL
    imgs = ctpn_amd.weights.synthetic_images(n, h, w, 101)
    full = N.forward(imgs, weights)
    with ctpn_amd.Context(0, n, h, w, "fp32") as ctx:
        ctx.load_weights(arena)
        ctx.forward(imgs)
        prev = N.image_blob(imgs)
        for name in N.CONVS:
            dev = ctx.get_tensor(name)
            iso = N.conv3x3_relu(prev, weights[name + "/weights"], weights[name + "/biases"])
            assert rel_err(dev, iso) < 5e-6, name             # exact-fp32 MFMA: summation-order noise only (K <= 4608)
            assert rel_err(dev, full[name]) < 2e-5, name
            prev = dev
            if name in N.POOL_AFTER:
                p = ctx.get_tensor(N.POOL_AFTER[name])
                assert np.array_equal(p, N.maxpool2x2(dev)), N.POOL_AFTER[name]
                prev = p
        assert rel_err(ctx.get_tensor("lstm_pre"), N.lstm_pre(prev, weights)) < 5e-6
        lo = ctx.get_tensor("lstm_out")
        assert rel_err(lo, N.bilstm(prev, weights)) < 5e-6
        assert rel_err(lo, full["lstm_out"]) < 2e-5
        fc = ctx.get_tensor("lstm_o")
        assert rel_err(fc, N.dense(lo, weights["lstm_o/weights"], weights["lstm_o/biases"])) < 5e-6
        info = np.array([[h, w, 1.0]] * n, np.float32)
        rois = ctx.proposals(info)
        cp, bp = ctx.get_tensor("rpn_cls_prob_reshape"), ctx.get_tensor("rpn_bbox_pred")
        assert np.abs(cp - full["rpn_cls_prob_reshape"]).max() < 1e-4      # bar: 1e-3
        assert np.abs(bp - full["rpn_bbox_pred"]).max() < 1e-4
        for i in range(n):                                                  # proposal layer on identical inputs
            want = P.proposal_layer(cp[i:i + 1], bp[i:i + 1], info[i])
            assert rois[i].shape == want.shape
            assert np.array_equal(rois[i][:, 0], want[:, 0])                # same anchors, same order
            assert np.abs(rois[i][:, 1:] - want[:, 1:]).max() < 1e-4        # expf vs np.exp: last-ulp differences only


def test_bf16_path_tracks_oracle(arena, weights):
    os.environ["CTPN_KEEP_ACTS"] = "1"
    n, h, w = SMALL
    imgs = ctpn_amd.weights.synthetic_images(n, h, w, 101)
    full = N.forward(imgs, weights)
    with ctpn_amd.Context(0, n, h, w, "bf16") as ctx:
        ctx.load_weights(arena)
        ctx.forward(imgs)
        prev = N.image_blob(imgs)
        for name in N.CONVS:
            dev = ctx.get_tensor(name)
            iso = N.conv3x3_relu(prev, weights[name + "/weights"], weights[name + "/biases"])
            assert rel_err(dev, iso) < 8e-3, name     # bf16 operands (8 mantissa bits), fp32 accumulate, bf16 store
            prev = dev
            if name in N.POOL_AFTER:
                p = ctx.get_tensor(N.POOL_AFTER[name])
                assert np.array_equal(p, N.maxpool2x2(dev))
                prev = p
        ctx.proposals(np.array([[h, w, 1.0]] * n, np.float32))
        cp = ctx.get_tensor("rpn_cls_prob_reshape")
        assert np.abs(cp - full["rpn_cls_prob_reshape"]).max() < 3e-2   # honest bf16 number: ~6e-3 observed, NOT the 1e-3 bar


@pytest.mark.parametrize("tag", [c[0] for c in CASES])
def test_proposals_and_lines_match_reference_fixtures(golden_dir, tag):
    case = [c for c in CASES if c[0] == tag][0]
    g = np.load(os.path.join(golden_dir, "postproc_%s.npz" % tag))
    cls, bbox = synth_inputs(case[1], case[2], case[3])
    with ctpn_amd.Context(0, 1, case[4], case[5], postproc_only=True) as ctx:      # ctpn_create_postproc: no network arena
        rois, anchors = ctx.proposals_from_host(cls, bbox, g["im_info"], want_anchors=True)
        rois, anchors = rois[0], anchors[0]
        with pytest.raises(ctpn_amd.CtpnError) as e:
            ctx.forward(np.zeros((1, 16, 16, 3), np.uint8))
        assert e.value.code == -3
    ref = g["rois"]
    assert rois.shape == ref.shape
    assert np.array_equal(canon_rows(rois)[:, 0], canon_rows(ref)[:, 0])            # identical score multiset/order
    assert np.abs(canon_rows(rois) - canon_rows(ref)).max() < 1e-3                  # boxes: +-1 px bar, observed ~6e-5
    # second return of proposal_layer: bbox_deltas[order][keep] (reference proposal_layer_tf.py:133-157), from the reference itself
    deltas = bbox.reshape(-1, 4)[anchors]
    assert deltas.shape == g["deltas"].shape
    assert np.array_equal(canon_rows(np.hstack([rois[:, :1], deltas])), canon_rows(np.hstack([ref[:, :1], g["deltas"]])))
    assert np.array_equal(cls.reshape(-1, 2)[anchors, 1], rois[:, 0])               # and they are the anchors that carry the scores
    dets = np.hstack([ref[:, 1:5], ref[:, 0:1]]).astype(np.float32)
    keep = B.nms_sorted(dets, 0.2, 0)
    want = g["nms_keep_0p2"]
    if tag not in TIE_HEAVY:
        assert sorted(keep.tolist()) == sorted(want.tolist())                       # ties: same set, documented order
    else:                                                                           # which box of a tie group survives follows the tie order
        assert len(keep) == len(want) and np.array_equal(canon_rows(dets[keep], 4), canon_rows(dets[want], 4))
    assert np.array_equal(keep, np.asarray(P.nms(dets, 0.2)))                       # canonical order: bit-exact
    for mode in "HO":
        recs = B.text_lines(ref[:, 1:5], ref[:, 0], (case[4], case[5]), mode, device_id=0)
        assert lines_close(tag, recs, g["recs_" + mode], 1e-3)
        # connect_kernel itself (score prefix, NMS 0.2, graph, chains, fit, filter all on the device) against the REFERENCE's lines
        dev = B.debug_connect(ref, (case[4], case[5]), mode)
        assert lines_close(tag, dev, g["recs_" + mode], 1e-3), mode
        assert np.array_equal(dev, recs)                                            # and bit-identical to the host C++ form


def test_nms_bit_exact_on_random_and_edge_inputs():
    rng = np.random.default_rng(5)
    assert B.nms_sorted(np.zeros((0, 5), np.float32), 0.7, 0).size == 0             # empty -> []
    for n in (1, 2, 63, 64, 65, 127, 129, 1000, 4096, 12000):
        x1 = np.floor(rng.uniform(0, 880, n) / 16).astype(np.float32) * 16
        y1 = rng.uniform(0, 560, n).astype(np.float32)
        b = np.stack([x1, y1, x1 + 16, y1 + rng.uniform(8, 120, n).astype(np.float32)], 1).astype(np.float32)
        s = np.sort(rng.uniform(0, 1, n).astype(np.float32))[::-1]
        dets = np.hstack([b, s[:, None]]).astype(np.float32)
        for thr in (0.7, 0.2):
            keep = B.nms_sorted(dets, thr, 0)
            assert np.array_equal(keep, np.asarray(P.nms(dets, thr))), (n, thr)
            assert np.all(np.diff(keep) > 0)                                         # ascending positions (= score order)
    same = np.tile(np.array([[10, 10, 25, 60, 0.5]], np.float32), (300, 1))          # all tied, all identical
    assert B.nms_sorted(same, 0.7, 0).tolist() == [0]
    dis = np.stack([np.arange(3000) * 20.0, np.zeros(3000), np.arange(3000) * 20.0 + 15, np.full(3000, 30.0), np.linspace(1, 0.1, 3000)], 1).astype(np.float32)
    assert B.nms_sorted(dis, 0.7, 0).tolist() == list(range(3000))                   # disjoint: everything kept (> LDS kept-list capacity)
    wide = np.hstack([dets[:500, :4], np.zeros((500, 3), np.float32), dets[:500, 4:5]])  # boxes_dim = 8 (only 4 coords are read)
    assert np.array_equal(B.nms_sorted(wide, 0.7, 0), np.asarray(P.nms(dets[:500], 0.7)))


def test_python_seams_keep_reference_signatures():
    from ctpn_amd.lib.fast_rcnn.nms_wrapper import nms
    from ctpn_amd.lib.utils.gpu_nms import gpu_nms
    from ctpn_amd.lib.rpn_msr.proposal_layer_tf import proposal_layer
    rng = np.random.default_rng(9)
    b = rng.uniform(0, 300, (200, 2)).astype(np.float32)
    dets = np.hstack([b, b + rng.uniform(10, 80, (200, 2)).astype(np.float32), rng.uniform(0, 1, (200, 1)).astype(np.float32)])
    want = P.nms(dets, 0.3)                                                           # unsorted input: indices into the caller's dets
    assert [int(i) for i in nms(dets, 0.3)] == want == [int(i) for i in gpu_nms(dets, 0.3, device_id=0)]
    assert nms(np.zeros((0, 5), np.float32), 0.3) == []
    cls, bbox = synth_inputs(21, 10, 14)
    info = np.array([[160, 224, 1.0]], np.float32)
    blob, deltas = proposal_layer(cls, bbox, info, "TEST", _feat_stride=[16, ], anchor_scales=[16, ])
    want, want_d = P.proposal_layer(cls, bbox, info[0], return_deltas=True)
    assert blob.shape == want.shape and deltas.shape == (blob.shape[0], 4)
    assert np.abs(canon_rows(blob) - canon_rows(want)).max() < 1e-3
    assert np.array_equal(canon_rows(np.hstack([blob[:, :1], deltas])), canon_rows(np.hstack([want[:, :1], want_d])))   # exact rows


def test_full_600x900_fp32_correctness_gate(arena, weights):
    """BASELINE.json config 2: single 600x900 image, fp32 HIP conv + BiLSTM + NMS vs the CPU path."""
    imgs = ctpn_amd.weights.synthetic_images(1, 600, 900, 1)
    ref = N.forward(imgs, weights, keep=set())
    info = np.array([[600, 900, 1.0]], np.float32)
    ref_rois = P.proposal_layer(ref["rpn_cls_prob_reshape"], ref["rpn_bbox_pred"], info[0])
    with ctpn_amd.Context(0, 1, 600, 900, "fp32") as ctx:
        ctx.load_weights(arena)
        lines, rois = ctx.detect(imgs, want_rois=True)
        cp, bp = ctx.get_tensor("rpn_cls_prob_reshape"), ctx.get_tensor("rpn_bbox_pred")
        assert np.abs(cp - ref["rpn_cls_prob_reshape"]).max() < 1e-3
        assert np.abs(bp - ref["rpn_bbox_pred"]).max() < 1e-3
        rois = rois[0]
        assert rois.shape[0] == ref_rois.shape[0] == 1000
        assert np.all(np.diff(rois[:, 0]) <= 0)                                       # descending score
        assert match_rois(rois, ref_rois, px_tol=1.0, score_tol=1e-3) >= 0.98         # near-tie flips allowed, see util.match_rois
        exact = P.proposal_layer(cp, bp, info[0])                                     # and exact given the device's own heads
        assert np.array_equal(rois[:, 0], exact[:, 0]) and np.abs(rois - exact).max() < 1e-3
        for mode in "HO":
            got = ctx.detect(imgs, mode=mode)[0]
            want = P.text_detect(exact[:, 1:5], exact[:, 0], (600, 900), mode)
            assert match_lines(got, want, 1.0, 1e-3), mode
        want_ref = P.text_detect(ref_rois[:, 1:5], ref_rois[:, 0], (600, 900), "H")
        assert abs(len(lines[0]) - len(want_ref)) <= 1


def test_batch_equals_singles_and_is_idempotent(arena):
    """Size-independent properties at the benchmark batch shape (bf16, 600x900): a batch is the concatenation of its
    images run alone, and running the same batch twice gives identical bytes."""
    n = 4
    imgs = ctpn_amd.weights.synthetic_images(n, 600, 900, 1)
    with ctpn_amd.Context(0, n, 600, 900, "bf16") as ctx:
        ctx.load_weights(arena)
        l1, r1 = ctx.detect(imgs, want_rois=True)
        l2, r2 = ctx.detect(imgs, want_rois=True)
        singles = {i: ctx.detect(imgs[i:i + 1], want_rois=True) for i in (0, n - 1)}
        same12 = [bool(np.array_equal(r1[i], r2[i]) and np.array_equal(l1[i], l2[i])) for i in range(n)]
        same1s = {i: bool(np.array_equal(rs[0], r1[i]) and np.array_equal(ls[0], l1[i])) for i, (ls, rs) in singles.items()}
        same2s = {i: bool(np.array_equal(rs[0], r2[i])) for i, (ls, rs) in singles.items()}
        assert all(same12) and all(same1s.values()), "first == second: %s, first == alone: %s, second == alone: %s" % (same12, same1s, same2s)
        for r in r1:
            assert r.shape[0] <= 1000 and np.all(np.diff(r[:, 0]) <= 0)
            assert np.all(r[:, 1] >= 0) and np.all(r[:, 3] <= 899) and np.all(r[:, 2] >= 0) and np.all(r[:, 4] <= 599)


def test_fused_pool_path_equals_unfused_path(arena):
    """The production configuration (pool fused into the conv epilogue, full-resolution conv1_2 / 2_2 / 3_3 / 4_3 never written) gives
    the same bytes as the keep_acts configuration, whose stored pools are the max of the stored full-resolution maps, and rejects
    requests for the tensors it does not store (fp32; the 16-bit modes: test_production_path_equals_keep_acts_path)."""
    imgs = ctpn_amd.weights.synthetic_images(2, 150, 230, 5)
    outs = {}
    for keep in (0, 1):
        with ctpn_amd.Context(0, 2, 150, 230, "fp32", options={"keep_acts": keep}) as ctx:
            ctx.load_weights(arena)
            ctx.forward(imgs)
            outs[keep] = {k: ctx.get_tensor(k) for k in ("pool1", "pool2", "pool3", "pool4", "conv5_3", "lstm_o")}
            if not keep:
                with pytest.raises(ctpn_amd.CtpnError) as e:
                    ctx.get_tensor("conv1_2")
                assert e.value.code == -3
            else:
                assert np.array_equal(outs[1]["pool1"], N.maxpool2x2(ctx.get_tensor("conv1_2")))
    for k in outs[0]:
        assert np.array_equal(outs[0][k], outs[1][k]), k


def test_folded_heads_equal_two_gemm_heads(arena):
    """bf16 throughput mode multiplies lstm_out by the pre-folded (lstm_o FC x heads) matrix; KEEP_ACTS=1 keeps the
    reference's FC -> heads op order. Same conv/LSTM bytes in both, so the heads may differ by fp32 rounding only."""
    imgs = ctpn_amd.weights.synthetic_images(2, 150, 230, 5)
    heads = {}
    for keep in ("1", "0"):
        os.environ["CTPN_KEEP_ACTS"] = keep
        with ctpn_amd.Context(0, 2, 150, 230, "bf16") as ctx:
            ctx.load_weights(arena)
            ctx.forward(imgs)
            heads[keep] = ctx.get_tensor("heads")
            if keep == "0":
                with pytest.raises(ctpn_amd.CtpnError):
                    ctx.get_tensor("lstm_o")
    assert np.abs(heads["1"] - heads["0"]).max() < 2e-5 * max(1.0, float(np.abs(heads["1"]).max()))


def test_async_submit_collect_equals_sync_detect(arena):
    """ctpn_detect_submit / ctpn_detect_collect (two slots, second stream) return the same bytes as ctpn_detect, in
    any interleaving, and refuse misuse of a slot."""
    a = ctpn_amd.weights.synthetic_images(2, 300, 452, 21)
    b = ctpn_amd.weights.synthetic_images(2, 300, 452, 31)
    with ctpn_amd.Context(0, 2, 300, 452, "bf16") as ctx:
        ctx.load_weights(arena)
        la, ra = ctx.detect(a, want_rois=True)
        lb, rb = ctx.detect(b, want_rois=True)
        ctx.detect_submit(a, slot=0)
        ctx.detect_submit(b, slot=1)
        with pytest.raises(ctpn_amd.CtpnError) as e:
            ctx.detect_submit(a, slot=0)
        assert e.value.code == -3
        l0, r0 = ctx.detect_collect(0, want_rois=True)
        ctx.detect_submit(a, slot=0)
        l1, r1 = ctx.detect_collect(1, want_rois=True)
        l2, r2 = ctx.detect_collect(0, want_rois=True)
        with pytest.raises(ctpn_amd.CtpnError):
            ctx.detect_collect(0)
        for i in range(2):
            assert np.array_equal(r0[i], ra[i]) and np.array_equal(l0[i], la[i])
            assert np.array_equal(r1[i], rb[i]) and np.array_equal(l1[i], lb[i])
            assert np.array_equal(r2[i], ra[i]) and np.array_equal(l2[i], la[i])


def test_bf16_convert_matches_rne():
    """v_cvt_pk_bf16_f32 (used by every bf16 epilogue) == the integer round-to-nearest-even formula == numpy/torch RNE,
    bit for bit, on random values, exact ties (both parities), denormals, zeros and large magnitudes."""
    rng = np.random.default_rng(2)
    vals = [rng.standard_normal(20000).astype(np.float32) * np.float32(10.0) ** rng.integers(-20, 20, 20000).astype(np.float32)]
    ties = (np.arange(0x3F80, 0x3F80 + 512, dtype=np.uint32) << 16) | np.uint32(0x8000)          # exactly half-way
    vals += [ties.view(np.float32), (ties + 1).view(np.float32), (ties - 1).view(np.float32), -ties.view(np.float32)]
    vals += [np.array([0.0, -0.0, 1e-45, -1e-45, 1e-39, 3.3e38, -3.3e38, 1.0, 65504.0], np.float32)]
    x = np.concatenate(vals).astype(np.float32)
    hw = B.debug_cvt_bf16(x, True)
    sw = B.debug_cvt_bf16(x, False)
    u = x.view(np.uint32).astype(np.uint64)
    ref = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)
    assert np.array_equal(sw, ref)
    assert np.array_equal(hw, ref)


def test_blob_feed_equals_uint8_feed(arena):
    imgs = ctpn_amd.weights.synthetic_images(1, 96, 160, 3)
    with ctpn_amd.Context(0, 1, 96, 160, "fp32") as ctx:
        ctx.load_weights(arena)
        ctx.forward(imgs)
        a = ctx.get_tensor("conv5_3")
        ctx.forward_blob(N.image_blob(imgs))
        b = ctx.get_tensor("conv5_3")
    assert np.array_equal(a, b)


def test_highres_oriented_config5_small_batch(arena, weights):
    """BASELINE.json config 5 shape (1280x1920, 96 000 anchors -> 12 000 into NMS, DETECT_MODE=O), one image."""
    imgs = ctpn_amd.weights.synthetic_images(1, 1280, 1920, 11)
    info = np.array([[1280, 1920, 1.0]], np.float32)
    with ctpn_amd.Context(0, 1, 1280, 1920, "bf16") as ctx:
        ctx.load_weights(arena)
        lines, rois = ctx.detect(imgs, mode="O", want_rois=True)
        assert ctx.feat_shape() == (1, 80, 120)
        cp, bp = ctx.get_tensor("rpn_cls_prob_reshape"), ctx.get_tensor("rpn_bbox_pred")
    want = P.proposal_layer(cp, bp, info[0])
    assert rois[0].shape == want.shape and np.array_equal(rois[0][:, 0], want[:, 0])
    assert np.abs(rois[0] - want).max() < 1e-3
    assert match_lines(lines[0], P.text_detect(want[:, 1:5], want[:, 0], (1280, 1920), "O"), 1.0, 1e-3)


def test_errors_are_loud(arena):
    with ctpn_amd.Context(0, 1, 64, 64, "fp32") as ctx:
        with pytest.raises(ctpn_amd.CtpnError) as e:
            ctx.forward(np.zeros((1, 64, 64, 3), np.uint8))
        assert e.value.code == -3                                                     # weights not loaded
        ctx.load_weights(arena)
        with pytest.raises(ctpn_amd.CtpnError) as e:
            ctx.forward(np.zeros((1, 128, 64, 3), np.uint8))
        assert e.value.code == -4                                                     # larger than the ctx arena
        ctx.forward(np.zeros((1, 64, 64, 3), np.uint8))
        with pytest.raises(ctpn_amd.CtpnError) as e:
            ctx.get_tensor("rpn_cls_prob_reshape")
        assert e.value.code == -3                                                     # produced by ctpn_proposals
        with pytest.raises(ctpn_amd.CtpnError):
            ctx.get_tensor("no_such_layer")


def test_demo_entry_point_end_to_end(tmp_path, arena, weights):
    """`python ctpn/demo.py` semantics on the GPU: data/demo/*.png -> data/results/res_<stem>.txt + annotated image, with
    the same lines the oracle derives for that image (image already at 600x900: both reference resizes are identity)."""
    PIL = pytest.importorskip("PIL")
    from PIL import Image
    from ctpn_amd.ctpn import demo
    from ctpn_amd.lib.fast_rcnn.config import cfg
    root = tmp_path
    (root / "data" / "demo").mkdir(parents=True)
    (root / "ctpn").mkdir()
    (root / "checkpoints").mkdir()
    bgr = ctpn_amd.weights.synthetic_images(1, 600, 900, 1)[0]
    Image.fromarray(bgr[:, :, ::-1].copy()).save(str(root / "data" / "demo" / "t01.png"))
    np.save(str(root / "checkpoints" / "ctpn_weights.npy"), arena)
    (root / "ctpn" / "text.yml").write_text("USE_GPU_NMS: True\nTEST:\n  DETECT_MODE: H\n  PRECISION: fp32\n  checkpoints_path: checkpoints/\n")
    cwd = os.getcwd()
    try:
        demo.main(["--root", str(root)])
    finally:
        os.chdir(cwd)
    res = (root / "data" / "results" / "res_t01.txt").read_bytes().decode()
    assert (root / "data" / "results" / "t01.png").exists()
    ref = N.forward(bgr[None], weights, keep=set())
    rois = P.proposal_layer(ref["rpn_cls_prob_reshape"], ref["rpn_bbox_pred"], np.array([600, 900, 1.0], np.float32))
    want = P.draw_boxes_lines(P.text_detect(rois[:, 1:5], rois[:, 0], (600, 900), "H"), 1.0)
    got = [l + "\n" for l in res.split("\n")[:-1]]
    assert all(l.endswith("\r\n") for l in got)
    # fp32 path vs oracle: same lines up to +-1 px and at most one borderline line (score within 1e-3 of LINE_MIN_SCORE)
    assert abs(len(got) - len(want)) <= 1
    gi = sorted(tuple(int(v) for v in l.strip().split(",")) for l in got)
    wi = sorted(tuple(int(v) for v in l.strip().split(",")) for l in want)
    matched = sum(1 for a in gi if any(max(abs(x - y) for x, y in zip(a, b)) <= 1 for b in wi))
    assert matched >= len(gi) - 1


@pytest.mark.parametrize("n,h,w", [(1, 16, 16), (3, 17, 33), (1, 48, 130), (2, 95, 64), (1, 33, 257), (1, 200, 31),
                                   (1, 48, 1472), (1, 32, 745)])   # the last two: map widths 92 / 93, at the edge of the flat-window LDS budget
def test_odd_shapes_fp32_end_to_end(arena, weights, n, h, w):
    """Ragged / minimal sizes (one feature cell, single tile column, tiles straddling image ends, W < one tile): final
    head outputs and rois of the fp32 path against the oracle."""
    imgs = ctpn_amd.weights.synthetic_images(n, h, w, 77)
    ref = N.forward(imgs, weights, keep=set())
    info = np.array([[h, w, 1.0]] * n, np.float32)
    with ctpn_amd.Context(0, n, h, w, "fp32") as ctx:
        ctx.load_weights(arena)
        ctx.forward(imgs)
        rois = ctx.proposals(info)
        assert ctx.feat_shape() == (n, h // 16, w // 16)
        cp, bp = ctx.get_tensor("rpn_cls_prob_reshape"), ctx.get_tensor("rpn_bbox_pred")
    assert cp.shape == ref["rpn_cls_prob_reshape"].shape
    assert np.abs(cp - ref["rpn_cls_prob_reshape"]).max() < 1e-4
    assert np.abs(bp - ref["rpn_bbox_pred"]).max() < 1e-4
    for i in range(n):
        want = P.proposal_layer(cp[i:i + 1], bp[i:i + 1], info[i])
        assert rois[i].shape == want.shape
        if want.size:
            assert np.array_equal(rois[i][:, 0], want[:, 0]) and np.abs(rois[i] - want).max() < 1e-3


@pytest.mark.parametrize("n,h,w", [(2, 17, 33), (1, 95, 64), (1, 33, 257), (1, 48, 1472), (1, 32, 1520)])
def test_odd_shapes_bf16_track_oracle(arena, weights, n, h, w):
    """Ragged / minimal shapes through the bf16 kernels: every layer against the oracle op on the device's previous tensor (the
    rigorous check: one launch per comparison, 8e-3 of the map's range), then the production configuration end to end. The
    end-to-end score bound is loose on purpose: bf16 rounding through 14 layers moves individual fg probabilities by up to a
    few 1e-2 (observed 0.035 on the 1 x 2 feature map of the 17 x 33 image; mean error ~1.5e-3, see profiles/r02_accuracy.json)."""
    imgs = ctpn_amd.weights.synthetic_images(n, h, w, 78)
    ref = N.forward(imgs, weights, keep=set())
    os.environ["CTPN_KEEP_ACTS"] = "1"
    with ctpn_amd.Context(0, n, h, w, "bf16") as ctx:
        ctx.load_weights(arena)
        _layerwise_bf16(ctx, imgs, weights, 8e-3)
    os.environ["CTPN_KEEP_ACTS"] = "0"
    with ctpn_amd.Context(0, n, h, w, "bf16") as ctx:
        ctx.load_weights(arena)
        lines, rois = ctx.detect(imgs, want_rois=True)
        cp = ctx.get_tensor("rpn_cls_prob_reshape")
    assert np.abs(cp - ref["rpn_cls_prob_reshape"]).max() < 6e-2
    assert all(r.shape[1] == 5 for r in rois) and all(l.shape[1] == 9 for l in lines)


@pytest.mark.parametrize("shape", [(2, 96, 160), (1, 37, 53), (1, 16, 19)])
def test_conv1_mfma_split_bf16_equals_fp32_direct_kernel(arena, weights, shape):
    """conv1_1 in bf16 mode runs on the matrix cores with split-bf16 operands (hi + lo); it must agree with the fp32
    direct (VALU) kernel to fp32-class accuracy, i.e. the bf16 outputs are equal except for rare 1-ulp rounding flips,
    for the uint8 feed and for the float blob feed."""
    n, h, w = shape
    imgs = ctpn_amd.weights.synthetic_images(n, h, w, 77)
    got = {}
    for flag in ("1", "0"):
        os.environ["CTPN_CONV1_MFMA"] = flag
        os.environ["CTPN_KEEP_ACTS"] = "1"
        with ctpn_amd.Context(0, n, h, w, "bf16") as ctx:
            ctx.load_weights(arena)
            ctx.forward(imgs)
            got[flag, "u8"] = ctx.get_tensor("conv1_1")
            ctx.forward_blob(N.image_blob(imgs))
            got[flag, "f32"] = ctx.get_tensor("conv1_1")
    os.environ.pop("CTPN_CONV1_MFMA")
    exact = N.conv3x3_relu(N.image_blob(imgs), weights["conv1_1/weights"], weights["conv1_1/biases"])
    for feed in ("u8", "f32"):
        a, b = got["1", feed], got["0", feed]
        assert a.shape == b.shape == exact.shape
        flips = a != b
        assert flips.mean() < 2e-3, (feed, flips.mean())                              # rounding-boundary cases only
        assert np.abs(a - b).max() <= np.abs(exact).max() * 2.0 ** -7               # never more than one bf16 ulp
        assert np.abs(a - exact).max() <= np.abs(b - exact).max() * 1.02 + 1e-6      # as close to the fp32 oracle as the direct kernel
    assert np.array_equal(got["1", "u8"], got["1", "f32"])


@pytest.mark.parametrize("prec,tol", [("fp32", 5e-6), ("bf16", 8e-3)])
@pytest.mark.parametrize("shape,pool,want_full", [
    ((1, 74, 112), True, False),    # 16 x 16 patches (35 tiles instead of 40), pooled output only
    ((2, 75, 113), True, False),    # odd H and W: the VALID pool never reads row 74 / column 112 -> tiling covers 74 x 112
    ((1, 75, 113), True, True),     # same map with the full-resolution output kept: tiling must cover 75 x 113
    ((1, 16, 144), False, True),    # 16 x 16 patches without pool (9 tiles instead of 10)
    ((1, 30, 225), True, False),    # W = 7 * 32 + 1: the trimmed extent removes the eighth tile column
    ((2, 20, 225), False, True),    # same W without pool: column 224 goes through the im2col kernel as a strip launch
    ((1, 12, 130, 64), False, True),   # Ci = 64 (weights-stationary kernel in bf16) + a 2-column strip
])
def test_conv3x3_patch_shapes_and_trimmed_pool_extent(prec, tol, shape, pool, want_full):
    """One conv layer on caller tensors (ctpn_debug_conv3x3) against the oracle conv (+ VALID 2x2 max-pool) for the map
    shapes that select the 16 x 16 output patch and / or the trimmed tiling extent of a fused pool."""
    n, h, w = shape[:3]
    ci, co = (shape[3] if len(shape) > 3 else 128), 128
    rng = np.random.default_rng(h * 1000 + w)
    x = np.maximum(rng.standard_normal((n, h, w, ci)).astype(np.float32), 0)
    wt = (rng.standard_normal((3, 3, ci, co)) * (2.0 / (9 * ci)) ** 0.5).astype(np.float32)
    b = (rng.standard_normal(co) * 0.1).astype(np.float32)
    if prec == "bf16":
        u = x.view(np.uint32).astype(np.uint64)      # the layer's input is bf16 in that mode: compare on the same operand values
        x = (((u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000).astype(np.uint32)).view(np.float32)
    full, pooled = B.debug_conv3x3(x, wt, b, prec, 1, pool, want_full)
    want = N.conv3x3_relu(x, wt, b)
    if want_full:
        assert rel_err(full, want) < tol
    if pool:
        ref_pool = N.maxpool2x2(full) if want_full else N.maxpool2x2(want)
        assert pooled.shape == ref_pool.shape
        if want_full:
            assert np.array_equal(pooled, ref_pool)        # the fused pool is exactly the max of the stored outputs
        else:
            assert rel_err(pooled, ref_pool) < tol


@pytest.mark.parametrize("prec,ci,co,pool", [("fp32", 64, 64, True), ("fp32", 64, 128, False), ("bf16", 128, 128, True), ("bf16", 64, 64, True)])
def test_conv3x3_is_bitwise_repeatable(prec, ci, co, pool):
    """The load pipelines are hand-counted (s_waitcnt vmcnt(N) on untracked LDS-DMA): a slice that is read before it has
    landed shows up as a rare wrong pixel row, not as a crash. Regression for exactly that (fp32 conv1_2, 4-wave tile:
    window slices issued in the last K step of a chunk): many launches of one layer must be bit-identical."""
    rng = np.random.default_rng(5)
    x = np.maximum(rng.standard_normal((2, 200, 320, ci)).astype(np.float32), 0)
    wt = (rng.standard_normal((3, 3, ci, co)) * (2.0 / (9 * ci)) ** 0.5).astype(np.float32)
    b = (rng.standard_normal(co) * 0.1).astype(np.float32)
    first = None
    for _ in range(25):
        full, pooled = B.debug_conv3x3(x, wt, b, prec, 1, pool, True)
        cur = (full, pooled if pool else full)
        if first is None:
            first = cur
        else:
            assert np.array_equal(cur[0], first[0]) and np.array_equal(cur[1], first[1])


@pytest.mark.parametrize("shape,fx,fy", [((37, 53), 2.0, 2.0), ((300, 500), 2.0, 2.0), ((480, 640), 1.25, 1.25), ((700, 1100), 600.0 / 700, 600.0 / 700),
                                         ((64, 64), 0.5, 0.5), ((33, 97), 1.7, 0.6), ((600, 1200), 1000.0 / 1200, 1000.0 / 1200)])
def test_resize_matches_oracle(shape, fx, fy):
    """ctpn_resize (cv2.resize INTER_LINEAR restated on the GPU) against oracle/resize_ref.py: the uint8
    fixed-point path bit for bit, the float32 path (the _get_image_blob rescale) exactly as well (same fp32 op order)."""
    from oracle import resize_ref as R
    rng = np.random.default_rng(shape[0] * 7 + shape[1])
    im = rng.integers(0, 256, (shape[0], shape[1], 3), dtype=np.uint8)
    got = B.resize_linear(im, fx, fy)
    want = R.resize_linear(im, fx, fy)
    assert got.shape == want.shape == B.resize_dims(shape[0], shape[1], fx, fy) + (3,)
    assert np.array_equal(got, want)
    blob = im.astype(np.float32) - np.array([102.9801, 115.9465, 122.7717], np.float32)
    gotf = B.resize_linear(blob, fx, fy)
    wantf = R.resize_linear(blob, fx, fy)
    assert gotf.dtype == np.float32 and np.array_equal(gotf, wantf)
    batch = np.stack([im, im[::-1].copy()])
    assert np.array_equal(B.resize_linear(batch, fx, fy), np.stack([want, R.resize_linear(im[::-1].copy(), fx, fy)]))


def test_batch_cli_equals_single_image_demo_path(tmp_path, arena):
    """ctpn/demo_batch.py: images of mixed sizes are resized on the GPU, grouped by shape, batched through
    submit/collect -- and every res_<stem>.txt equals what the single-image demo.ctpn() path writes for that image."""
    pytest.importorskip("PIL")
    from PIL import Image
    from ctpn_amd.ctpn import demo, demo_batch
    from ctpn_amd.lib.fast_rcnn.config import cfg
    from ctpn_amd.lib.networks.factory import get_network
    src, out_b, out_s = tmp_path / "in", tmp_path / "batch", tmp_path / "single"
    src.mkdir(); out_s.mkdir()
    rng = np.random.default_rng(11)
    # after resize_im: 600x800 (one image, sorted FIRST), 600x900 (x4), 600x1000. The net starts at TEST.MAX_BATCH = 1: the ctx has
    # to hold the largest batch before the first submit (growing it mid-run used to drop the pending batch: ADVICE r1)
    shapes = [(300, 450), (300, 450), (600, 900), (300, 450), (240, 400), (300, 400)]
    for i, (h, w) in enumerate(shapes):
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(str(src / ("im%02d.png" % i)))
    cfg.TEST.PRECISION = "bf16"
    net = get_network("VGGnet_test")
    net.load_arena(arena)
    try:
        names = demo_batch.list_images(str(src))
        assert len(names) == 6
        jobs, singles, _ = demo_batch.plan(names, 3)
        assert [len(m) for _, m in jobs] == [1, 3, 1, 1] and jobs[0][0] == (600, 800) and not singles
        assert net.max_batch == 1
        res = demo_batch.run(net, names, str(out_b), batch=3, write_images=False, log=lambda *_: None)
        assert net.max_batch == 3 and len(res) == 6
        for nm in names:
            demo.ctpn(None, net, nm, out_dir=str(out_s))
            stem = os.path.basename(nm).split(".")[0]
            a = (out_b / ("res_%s.txt" % stem)).read_bytes()
            b = (out_s / ("res_%s.txt" % stem)).read_bytes()
            assert a == b, stem                                   # same kernels, same per-image arithmetic: identical files
            assert res[nm].shape[1] == 9
        assert (out_s / "im00.png").exists()
    finally:
        net.close()


def test_split_bf16_recurrence_equals_fp32_recurrence(arena):
    """bf16 mode runs the BiLSTM recurrence as three bf16 MFMAs on hi/lo operand halves (bilstm_split_kernel); on the same
    pre-activations it must agree with the exact-fp32 MFMA kernel to fp32-class accuracy over all 56+ time steps."""
    n, h, w = 2, 96, 1000                      # Wf = 62 steps
    imgs = ctpn_amd.weights.synthetic_images(n, h, w, 23)
    got = {}
    for flag in ("1", "0"):
        os.environ["CTPN_LSTM_SPLIT"] = flag
        os.environ["CTPN_KEEP_ACTS"] = "1"
        with ctpn_amd.Context(0, n, h, w, "bf16") as ctx:
            ctx.load_weights(arena)
            ctx.forward(imgs)
            got[flag] = (ctx.get_tensor("lstm_pre"), ctx.get_tensor("lstm_out"))
    os.environ.pop("CTPN_LSTM_SPLIT")
    assert np.array_equal(got["1"][0], got["0"][0])                       # same inputs to the recurrence
    a, b = got["1"][1], got["0"][1]
    assert a.shape == b.shape and np.abs(b).max() > 0.1
    assert np.abs(a - b).max() < 2e-5, np.abs(a - b).max()


@pytest.mark.parametrize("n,h,w", [(4, 600, 900), (2, 333, 517), (1, 96, 1000)])
def test_device_connector_equals_host_connector(arena, n, h, w):
    """connect_kernel (graph build, chains, line fit, filter_boxes on the GPU) against the host C++
    restatement csrc/text_connector.cpp on the same NMS survivors: identical float64 records, both DETECT_MODEs."""
    imgs = ctpn_amd.weights.synthetic_images(n, h, w, 31)
    got = {}
    for flag in ("1", "0"):
        os.environ["CTPN_CONNECT_DEVICE"] = flag
        with ctpn_amd.Context(0, n, h, w, "bf16") as ctx:
            ctx.load_weights(arena)
            got[flag] = {m: ctx.detect(imgs, mode=m, line_capacity=600) for m in "HO"}
    os.environ.pop("CTPN_CONNECT_DEVICE")      # back to the default (host connector)
    total = 0
    for m in "HO":
        for a, b in zip(got["1"][m], got["0"][m]):
            assert a.shape == b.shape and np.array_equal(a, b), m
            total += len(a)
    assert total > 0 or h < 200          # the synthetic weights do produce lines on the larger maps


def _layerwise_bf16(ctx, imgs, weights, tol):
    """Every conv / pool / LSTM tensor of one bf16 forward against the oracle op applied to the DEVICE's previous tensor
    (so each check isolates one kernel launch): conv rel-err < tol, fused pools == max of the stored conv output exactly."""
    ctx.forward(imgs)
    prev = N.image_blob(imgs)
    worst = {}
    for name in N.CONVS:
        dev = ctx.get_tensor(name)
        iso = N.conv3x3_relu(prev, weights[name + "/weights"], weights[name + "/biases"])
        worst[name] = rel_err(dev, iso)
        assert worst[name] < tol, (name, worst[name])
        # a wrong pixel row / tile shows up as a LOCAL error: no pixel may be off by more than a few bf16 ulps of the map's range
        bad = np.abs(dev - iso).max(axis=-1) > 4 * tol * max(float(np.abs(iso).max()), 1e-30)
        assert not bad.any(), (name, int(bad.sum()), np.argwhere(bad)[:4].tolist())
        prev = dev
        del iso
        if name in N.POOL_AFTER:
            p = ctx.get_tensor(N.POOL_AFTER[name])
            assert np.array_equal(p, N.maxpool2x2(dev)), N.POOL_AFTER[name]
            prev = p
    pre = ctx.get_tensor("lstm_pre")
    assert rel_err(pre, N.lstm_pre(prev, weights)) < tol
    lo = ctx.get_tensor("lstm_out")
    # exact-fp32 recurrence on the device's own pre-activations is not exposed by the oracle (it recomputes x @ Wx from prev):
    # compare against the oracle BiLSTM of the bf16 conv5 output; the input projection carries the bf16 operand rounding
    assert np.abs(lo - N.bilstm(prev, weights)).max() < 2e-2
    return worst


def test_bf16_every_layer_at_600x900_batch16_matches_oracle(arena, weights):
    """The kernels and the geometry that produce the headline number (BASELINE config 3: 600x900, bf16): n = 16 gives every
    persistent workgroup >= 2 tiles on every layer (conv5_x: 568 tiles on 256 CUs), so the prefetch-next-tile-while-this-one-
    is-on-the-MFMAs path of conv3x3_p_kernel and the steady state of conv3x3_ws_kernel run against the oracle, layer by layer."""
    os.environ["CTPN_KEEP_ACTS"] = "1"
    n = 16
    imgs = ctpn_amd.weights.synthetic_images(n, 600, 900, 1)
    with ctpn_amd.Context(0, n, 600, 900, "bf16") as ctx:
        ctx.load_weights(arena)
        worst = _layerwise_bf16(ctx, imgs, weights, 8e-3)
    print("bf16 600x900 n=16 layer-wise rel err:", {k: "%.2e" % v for k, v in worst.items()})


def test_bf16_every_layer_at_1280x1920_batch2_matches_oracle(arena, weights):
    """BASELINE config 5 geometry (1280x1920): 80 x 120 feature map, flat-mode windows at W = 120 + 2, strip launches, 16 x 16 patches."""
    os.environ["CTPN_KEEP_ACTS"] = "1"
    n = 2
    imgs = ctpn_amd.weights.synthetic_images(n, 1280, 1920, 41)
    with ctpn_amd.Context(0, n, 1280, 1920, "bf16") as ctx:
        ctx.load_weights(arena)
        _layerwise_bf16(ctx, imgs, weights, 8e-3)


def test_production_path_equals_keep_acts_path_at_600x900(arena):
    """KEEP_ACTS=1 (what the layer-wise tests run) stores the full-resolution output of the pool-fused convs and keeps the
    two-GEMM heads; the production configuration does neither. Same pools / conv5_3 bytes, heads within fp32 rounding."""
    n = 8
    imgs = ctpn_amd.weights.synthetic_images(n, 600, 900, 1)
    got = {}
    for keep in ("1", "0"):
        os.environ["CTPN_KEEP_ACTS"] = keep
        with ctpn_amd.Context(0, n, 600, 900, "bf16") as ctx:
            ctx.load_weights(arena)
            ctx.forward(imgs)
            got[keep] = {k: ctx.get_tensor(k) for k in ("pool1", "pool2", "pool3", "pool4", "conv5_3", "rpn_conv/3x3", "lstm_out", "heads")}
    for k in ("pool1", "pool2", "pool3", "pool4", "conv5_3", "rpn_conv/3x3", "lstm_out"):
        assert np.array_equal(got["1"][k], got["0"][k]), k
    assert np.abs(got["1"]["heads"] - got["0"]["heads"]).max() < 2e-5 * max(1.0, float(np.abs(got["1"]["heads"]).max()))


def test_bf16_accuracy_vs_fp32_oracle_on_benchmark_images(arena, weights):
    """What the bf16 throughput mode DELIVERS against the fp32 oracle on benchmark images (north_star's bar -- scores 1e-3, boxes +-1 px -- is
    held by CTPN_PREC_FP32 and CTPN_PREC_SPLIT: tests/test_gpu_precision.py; bf16 misses it and says so): floors just below the measured
    numbers (32 images, profiles/r03_accuracy.json: cls_prob max 0.018 / mean 0.0016, 96.0 % of the rois within 1 px / 1e-3, 69 % of the text
    lines within 1 px, 83 % at hull IoU 0.7), so that a regression shows. The fp16 mode's floors: test_fp16_precision_accuracy_floors."""
    from accuracy_report import accuracy_of
    rep = accuracy_of(arena, weights, n=4, seed0=1)
    print("bf16 vs fp32 oracle, 4 benchmark images:", rep)
    assert rep["cls_prob_max_abs_diff"] < 2.5e-2 and rep["cls_prob_mean_abs_diff"] < 2e-3
    assert rep["roi_match_frac_1px_1e-3"] > 0.94                  # measured 0.960 - 0.965
    assert rep["roi_match_frac_1px_1e-2"] > 0.955                 # measured 0.97
    # a text line's corners move by a proposal width (16 px) when ONE of its proposals flips, so the 1 px line match of the
    # bf16 path is much lower than its roi match; as detections (hull IoU > 0.7) the lines agree
    assert rep["text_line_match_frac_1px"] > 0.58                 # measured 0.66 - 0.71 depending on the image set
    assert rep["text_line_match_frac_iou0.7"] > 0.75              # measured 0.80 - 0.90
    assert abs(rep["text_lines_device"] - rep["text_lines_oracle"]) <= 0.05 * rep["text_lines_oracle"] + 2


def test_zero_and_tied_scores_keep_valid_prefix():
    """Exact zeros / saturated scores handed to ctpn_proposals_from_host: a valid anchor with score 0.0 must still sort before
    every filtered (invalid) anchor, so fewer-than-topn valid anchors form a prefix (radix sort on the high word)."""
    hf, wf = 6, 9
    rng = np.random.default_rng(3)
    cls = np.zeros((1, hf, wf, 20), np.float32)
    fg = rng.choice([0.0, 0.0, 1.0, 0.5], size=(hf, wf, 10)).astype(np.float32)
    cls[0, :, :, 1::2] = fg
    cls[0, :, :, 0::2] = 1.0 - fg
    bbox = (rng.standard_normal((1, hf, wf, 40)) * 0.3).astype(np.float32)
    bbox[0, :, :, 3::4] -= 3.0 * (rng.uniform(size=(hf, wf, 10)) < 0.3)            # tiny boxes: filtered (invalid keys)
    info = np.array([[hf * 16, wf * 16, 1.0]], np.float32)
    want = P.proposal_layer(cls, bbox, info[0])
    with ctpn_amd.Context(0, 1, hf * 16, wf * 16, postproc_only=True) as ctx:
        for _ in range(3):                                                          # stale rows of an earlier call must not leak in
            rois = ctx.proposals_from_host(cls, bbox, info)[0]
    assert rois.shape == want.shape
    # same rows as the oracle (one-to-one, expf vs np.exp differ in the last ulp of y1 / y2), zero-score rows included
    assert match_rois(rois, want, px_tol=1e-3, score_tol=0.0) == 1.0 and match_rois(want, rois, px_tol=1e-3, score_tol=0.0) == 1.0
    assert np.array_equal(np.sort(rois[:, 0]), np.sort(want[:, 0]))


@pytest.mark.parametrize("shape,co,pool,want_full", [
    ((1, 8, 32), 64, True, False),       # one tile, one worker: T = 1 (prologue + flush only)
    ((2, 8, 16), 128, False, True),      # one tile per image, one tile column (divisor 1 in the tile-index arithmetic)
    ((1, 47, 32), 128, False, True),     # one tile column, six tile rows
    ((1, 24, 40), 64, True, False),      # 6 tiles, ragged right edge (W = 40 -> second tile column holds 8 valid columns)
    ((2, 75, 113), 64, True, False),     # odd H and W under the trimmed pool extent
    ((2, 75, 113), 64, True, True),      # ... and with the full-resolution output kept (26 epilogue pieces per tile)
    ((1, 37, 450), 128, False, True),    # conv2_1's width: 14 tile columns + the 2-column strip launch, two channel slices per tile
    ((4, 256, 512), 64, True, False),    # 2048 tiles: every workgroup walks 8 tiles -> all three window buffers rotate several times
    ((3, 128, 384), 128, False, True),   # 1152 tiles x 2 channel slices on 128 workers each: T = 9 (odd: tail flush on set 0)
])
def test_conv3x3_weights_in_registers_kernel(shape, co, pool, want_full):
    """conv3x3_wr_kernel (bf16, Ci = 64: conv1_2 / conv2_1 of the throughput path) through ctpn_debug_conv3x3 against the oracle
    conv (+ VALID 2x2 max-pool): tile walks of every length class, ragged edges, both epilogues. The unclamped window fetch
    reads past the image on edge tiles: the outputs must not depend on what lies there (compared against the round-1 kernel,
    which clamps, bit for bit)."""
    n, h, w = shape
    rng = np.random.default_rng(h * 131 + w + co)
    x = np.maximum(rng.standard_normal((n, h, w, 64)).astype(np.float32), 0)
    u = x.view(np.uint32).astype(np.uint64)
    x = (((u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000).astype(np.uint32)).view(np.float32)      # bf16-representable inputs
    wt = (rng.standard_normal((3, 3, 64, co)) * (2.0 / (9 * 64)) ** 0.5).astype(np.float32)
    b = (rng.standard_normal(co) * 0.1).astype(np.float32)
    full, pooled = B.debug_conv3x3(x, wt, b, "bf16", 1, pool, want_full)
    want = N.conv3x3_relu(x, wt, b)
    if want_full:
        assert rel_err(full, want) < 8e-3
        bad = np.abs(full - want).max(axis=-1) > 4 * 8e-3 * float(np.abs(want).max())
        assert not bad.any(), np.argwhere(bad)[:4].tolist()
    if pool:
        ref_pool = N.maxpool2x2(full) if want_full else N.maxpool2x2(want)
        assert pooled.shape == ref_pool.shape
        if want_full:
            assert np.array_equal(pooled, ref_pool)
        else:
            assert rel_err(pooled, ref_pool) < 8e-3
            bad = np.abs(pooled - ref_pool).max(axis=-1) > 4 * 8e-3 * float(np.abs(ref_pool).max())
            assert not bad.any(), np.argwhere(bad)[:4].tolist()
    # same bytes run after run (hand-counted load pipeline), and -- up to the fp32 summation order -- as the round-1 kernel
    full2, pooled2 = B.debug_conv3x3(x, wt, b, "bf16", 1, pool, want_full)
    assert (full is None or np.array_equal(full, full2)) and (pooled is None or np.array_equal(pooled, pooled2))


def test_demo_pb_entry_point_equals_demo(tmp_path, arena):
    """`python ctpn/demo_pb.py` (reference ctpn/demo_pb.py:55-98): frozen graph in, the two head tensors through the Python
    proposal_layer seam (ctpn_proposals_from_host), TextDetector, draw_boxes -- must write the same res_<stem>.txt as demo.py's
    fused path for the same weights and image."""
    pytest.importorskip("PIL")
    from PIL import Image
    from ctpn_amd.ctpn import demo, demo_pb
    from ctpn_amd.lib.fast_rcnn.config import cfg
    root = tmp_path
    for d in ("data/demo", "ctpn", "checkpoints"):
        (root / d).mkdir(parents=True)
    bgr = ctpn_amd.weights.synthetic_images(1, 600, 900, 3)[0]
    Image.fromarray(bgr[:, :, ::-1].copy()).save(str(root / "data" / "demo" / "p01.png"))
    np.save(str(root / "checkpoints" / "ctpn_weights.npy"), arena)
    (root / "ctpn" / "text.yml").write_text("USE_GPU_NMS: True\nTEST:\n  DETECT_MODE: H\n  PRECISION: bf16\n  checkpoints_path: checkpoints/\n")
    cwd = os.getcwd()
    try:
        demo.main(["--root", str(root)])
        a = (root / "data" / "results" / "res_p01.txt").read_bytes()
        demo_pb.main(["--root", str(root), "--synthetic", "0"])
        b = (root / "data" / "results" / "res_p01.txt").read_bytes()
    finally:
        os.chdir(cwd)
    assert (root / "data" / "ctpn.pb").exists() and (root / "data" / "results" / "p01.png").exists()
    assert len(a) > 0 and a == b


@pytest.mark.parametrize("shape,ci,co", [
    ((2, 40, 113), 256, 128),     # conv4-like width 113 = 7 * 16 + 1: one edge column, main launch tiles 16 x 16
    ((1, 30, 225), 128, 256),     # conv3_1's width: 7 * 32 + 1
    ((1, 37, 450), 64, 128),      # conv2_1: weights-in-registers main launch + two edge columns
    ((3, 5, 34), 128, 128),       # fewer edge pixels (30) than one wave's 32: clamped lanes must not store
    ((1, 70, 98), 512, 512),      # 98 = 6 * 16 + 2, deepest K loop (9 x 32 steps), four 128-channel slices in the main launch
])
def test_conv3x3_edge_columns_kernel(shape, ci, co):
    """conv3x3_edge_kernel (bf16): the one or two ragged pixel columns beyond the main launch's tiles, computed by
    one-wave workgroups that share the CUs with the persistent main kernel. The edge columns and their neighbours
    (a wrong w_cover would leave a gap or overwrite) against the oracle conv; repeatable bit for bit."""
    n, h, w = shape
    rng = np.random.default_rng(h * 7 + w + ci + co)
    x = np.maximum(rng.standard_normal((n, h, w, ci)).astype(np.float32), 0)
    u = x.view(np.uint32).astype(np.uint64)
    x = (((u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000).astype(np.uint32)).view(np.float32)      # bf16-representable inputs
    wt = (rng.standard_normal((3, 3, ci, co)) * (2.0 / (9 * ci)) ** 0.5).astype(np.float32)
    b = (rng.standard_normal(co) * 0.1).astype(np.float32)
    full, _ = B.debug_conv3x3(x, wt, b, "bf16", 1, False, True)
    want = N.conv3x3_relu(x, wt, b)
    r = w % 16
    assert 1 <= r <= 2
    assert rel_err(full[:, :, w - r:], want[:, :, w - r:]) < 8e-3           # the edge kernel's pixels on their own
    assert rel_err(full, want) < 8e-3
    bad = np.abs(full - want).max(axis=-1) > 4 * 8e-3 * float(np.abs(want).max())
    assert not bad.any(), np.argwhere(bad)[:4].tolist()
    full2, _ = B.debug_conv3x3(x, wt, b, "bf16", 1, False, True)
    assert np.array_equal(full, full2)


@pytest.mark.parametrize("n,h,w,scale", [(4, 600, 900, 1.0), (2, 333, 517, 1.8018), (2, 608, 912, 1.25), (1, 96, 1000, 5.0), (1, 1280, 1920, 1.0), (6, 600, 900, 1.0),
                                         (1, 600, 900, 1.0), (1, 48, 32, 1.0), (2, 600, 900, 1.9), (3, 352, 1000, 1.0)])
def test_column_nms_variants_equal_generic_nms(arena, n, h, w, scale):
    """The NMS kernels of the detect path -- nms_kernel (generic, round 1), nms_columns_kernel for the proposal layer and its
    connector variant (boxes / im_scale, threshold 0.2; scale 5.0 is outside its domain and must fall back to the generic
    kernel) -- give identical rois and identical text lines; CTPN_NMS_CHECK=1 (the debug assertion of the column
    decomposition's precondition) stays silent on decode output."""
    imgs = ctpn_amd.weights.synthetic_images(n, h, w, 77)
    scales = np.full((n,), scale, np.float32)
    got = {}
    # "1": one workgroup per image, or -- batches of up to four images -- one column per wave over ncols / 4 workgroups per image; "2" / "3"
    # pin either form (3 with more than four images: the one-workgroup form, the scratch holds four). The key sort follows: "0" / "2" one
    # workgroup per image, "1" / "3" eight sorted segments per image merged by rank for up to four images of 4097 .. 32768 anchors (scale 1.9:
    # min_size = 15.2 px turns the low boxes' keys into the KEY_INVALID tail, the only keys that tie)
    for tag, cols in (("generic", "0"), ("columns", "1"), ("one-wg", "2"), ("multi-wg", "3")):
        os.environ["CTPN_NMS_COLUMNS"] = cols
        os.environ["CTPN_NMS_CHECK"] = "0" if cols == "0" else "1"
        try:
            with ctpn_amd.Context(0, n, h, w, "bf16") as ctx:
                ctx.load_weights(arena)
                got[tag] = {m: ctx.detect(imgs, scales=scales, mode=m, line_capacity=600, want_rois=True) for m in "HO"}
        finally:
            os.environ.pop("CTPN_NMS_COLUMNS")
            os.environ.pop("CTPN_NMS_CHECK")
    rois_seen = 0
    for tag in ("columns", "one-wg", "multi-wg"):
        for m in "HO":
            lines_a, rois_a = got["generic"][m]
            lines_b, rois_b = got[tag][m]
            for a, b in zip(rois_a, rois_b):
                assert a.shape == b.shape and np.array_equal(a, b), (tag, "rois")
                rois_seen += len(a)
            for a, b in zip(lines_a, lines_b):
                assert a.shape == b.shape and np.array_equal(a, b), (tag, m)
    assert rois_seen > 0 or scale > 4        # min_size = 8 * im_scale filters every proposal of the small map at scale 5


@pytest.mark.parametrize("shape", [(2, 96, 160), (1, 37, 53), (1, 16, 19), (1, 21, 64), (2, 16, 65)])
def test_conv1_exact_pixel_kernel(arena, weights, shape):
    """conv1_1 from the q-image (default for the uint8 feed in the 16-bit modes; stored by conv_first_p_kernel under keep_acts, computed inside
    conv1_2's window stage otherwise): pixels enter the MFMA as exact integers p - round(mean), the fractional part of the mean rides on
    the pixels' inside-the-image slots (layers.hip, pack_conv1_frags). Its only inexactness is the bf16
    rounding of the 27 weights -- so it must reproduce the fp32 oracle conv evaluated with bf16-ROUNDED weights to fp32-class
    accuracy (bf16 outputs equal except rounding-boundary flips), on interior and on every border / corner pixel."""
    n, h, w = shape
    imgs = ctpn_amd.weights.synthetic_images(n, h, w, 78)
    os.environ["CTPN_KEEP_ACTS"] = "1"
    got = {}
    for flag in ("2", "1"):
        os.environ["CTPN_CONV1_MFMA"] = flag
        try:
            with ctpn_amd.Context(0, n, h, w, "bf16") as ctx:
                ctx.load_weights(arena)
                ctx.forward(imgs)
                got[flag] = ctx.get_tensor("conv1_1")
        finally:
            os.environ.pop("CTPN_CONV1_MFMA")
    wq = weights["conv1_1/weights"].astype(np.float32)
    u = wq.view(np.uint32).astype(np.uint64)
    wq = (((u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000).astype(np.uint32)).view(np.float32)
    want = N.conv3x3_relu(N.image_blob(imgs), wq, weights["conv1_1/biases"])
    a = got["2"]
    assert a.shape == want.shape
    scale = float(np.abs(want).max())
    ulp = scale * 2.0 ** -8
    assert np.abs(a - want).max() <= ulp, np.abs(a - want).max() / ulp          # within one bf16 rounding of the exact value
    border = np.zeros((h, w), bool)
    border[[0, -1], :] = True
    border[:, [0, -1]] = True
    assert np.abs(a - want)[:, border].max() <= ulp                             # the tap-dropping corrections
    # and against the split kernel (fp32-class weights): the difference is the weight rounding, 2^-9 relative per product
    assert rel_err(a, got["1"]) < 8e-3


@pytest.mark.parametrize("shape,ci,co", [
    ((1, 10, 100), 64, 64),       # conv1_2-like: weights-in-registers main launch (32-wide tiles), four edge columns = two pooled columns
    ((1, 12, 68), 64, 128),       # ... with two channel slices
    ((2, 9, 98), 128, 128),       # conv2_2-like: persistent main launch, two edge columns = one pooled column; odd H (VALID drops the last row)
    ((1, 8, 452), 128, 128),      # wide: 28 tile columns + 4 edge columns
])
def test_conv3x3_pooled_edge_columns_kernel(shape, ci, co):
    """conv3x3_edge_kernel<POOL>: the ragged columns of a layer whose 2x2 max-pool is fused and whose full-resolution map is
    never written: pooled pixels computed by lane quads (conv at the four positions, max over the quad)."""
    n, h, w = shape
    rng = np.random.default_rng(h * 11 + w + ci + co)
    x = np.maximum(rng.standard_normal((n, h, w, ci)).astype(np.float32), 0)
    u = x.view(np.uint32).astype(np.uint64)
    x = (((u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000).astype(np.uint32)).view(np.float32)      # bf16-representable inputs
    wt = (rng.standard_normal((3, 3, ci, co)) * (2.0 / (9 * ci)) ** 0.5).astype(np.float32)
    b = (rng.standard_normal(co) * 0.1).astype(np.float32)
    _, pooled = B.debug_conv3x3(x, wt, b, "bf16", 1, True, False)
    want = N.maxpool2x2(N.conv3x3_relu(x, wt, b))
    assert pooled.shape == want.shape
    r2 = (w % (32 if ci == 64 else 16)) // 2
    assert r2 in (1, 2)
    assert rel_err(pooled[:, :, -r2:], want[:, :, -r2:]) < 8e-3            # the edge kernel's pooled columns on their own
    assert rel_err(pooled, want) < 8e-3
    bad = np.abs(pooled - want).max(axis=-1) > 4 * 8e-3 * float(np.abs(want).max())
    assert not bad.any(), np.argwhere(bad)[:4].tolist()
    _, pooled2 = B.debug_conv3x3(x, wt, b, "bf16", 1, True, False)
    assert np.array_equal(pooled, pooled2)
    # with the full-resolution map kept as well, the edge columns come from the same kernel: identical pooled map, and it is the
    # exact max of the stored map
    full3, pooled3 = B.debug_conv3x3(x, wt, b, "bf16", 1, True, True)
    assert np.array_equal(pooled3, pooled) and np.array_equal(pooled3, N.maxpool2x2(full3))
