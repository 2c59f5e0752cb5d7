"""GPU (-m gpu), round 5 (VERDICT r4 "next" 1, 3, 6): what the drop-in does out of the box, and the configurations that had no end-to-end
parity test.

  * `python ctpn/demo.py` with the SHIPPED ctpn/text.yml (cfg.TEST.PRECISION default "split") writes the same res_<stem>.txt bytes as the
    fp32 oracle path does for the same images;
  * BASELINE.json configs[4]'s geometry (1280 x 1920, DETECT_MODE = O) end to end, fp32 and split, against oracle/network.py +
    oracle/postproc.py with the 600 x 900 gate's assertions;
  * a lone image through the synchronous seams (ctpn_detect; test_ctpn + TextDetector.detect) gives the batch path's bytes;
  * eight ranks of bench.py on one device.
Nothing here reads /root/reference.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import ctpn_amd
from oracle import network as N
from oracle import postproc as P
from util import match_lines, match_rois

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def weights(arena):
    return ctpn_amd.arena_views(arena)


def test_demo_with_the_shipped_config_writes_the_oracle_paths_result_files(tmp_path, arena, weights):
    """ctpn/demo.py --synthetic 0, no ctpn/text.yml under --root, so the package's own text.yml and config defaults decide the arithmetic
    (PRECISION: split). Five synthetic images already at 600 x 900 (both reference resizes are identity): every res_<stem>.txt equals, byte for
    byte, what oracle/network.py (fp32) -> proposal_layer -> TextDetector -> draw_boxes' text writes for that image."""
    pytest.importorskip("PIL")
    from PIL import Image
    from ctpn_amd.ctpn import demo
    from ctpn_amd.lib.fast_rcnn.config import cfg
    assert cfg.TEST.PRECISION == "split"
    root = tmp_path
    (root / "data" / "demo").mkdir(parents=True)
    seeds = [1, 2, 3, 4, 5]
    imgs = {}
    for s in seeds:
        bgr = ctpn_amd.weights.synthetic_images(1, 600, 900, s)[0]
        imgs["s%02d" % s] = bgr
        Image.fromarray(bgr[:, :, ::-1].copy()).save(str(root / "data" / "demo" / ("s%02d.png" % s)))
    cwd = os.getcwd()
    try:
        demo.main(["--root", str(root), "--synthetic", "0"])
        assert cfg.TEST.PRECISION == "split" and cfg.TEST.DETECT_MODE == "H"       # what the shipped yml says
    finally:
        os.chdir(cwd)
    info = np.array([600, 900, 1.0], np.float32)
    n_lines = 0
    for stem, bgr in imgs.items():
        got = (root / "data" / "results" / ("res_%s.txt" % stem)).read_bytes()
        ref = N.forward(bgr[None], weights, keep=set())
        rois = P.proposal_layer(ref["rpn_cls_prob_reshape"], ref["rpn_bbox_pred"], info)
        want = "".join(P.draw_boxes_lines(P.text_detect(rois[:, 1:5], rois[:, 0], (600, 900), "H"), 1.0)).encode()
        assert got == want, "%s: the out-of-box demo's result file differs from the fp32 oracle path's" % stem
        assert (root / "data" / "results" / (stem + ".png")).exists()
        n_lines += got.count(b"\r\n")
    assert n_lines >= 20        # the comparison is not vacuous
    print("demo.py (shipped config, split precision): %d result lines over %d images byte-equal to the oracle path's" % (n_lines, len(seeds)))


@pytest.mark.parametrize("prec", ["fp32", "split"])
def test_config5_geometry_end_to_end_against_the_oracle(arena, weights, prec):
    """1 x 1280 x 1920, DETECT_MODE = O (80 x 120 feature map, 96 000 anchors, 12 000 into the NMS): the 600 x 900 gate's assertions
    (test_split_precision_holds_north_star_tolerance_at_batch_8) against the fp32 oracle forward + the reference-pinned post-processing."""
    h, w = 1280, 1920
    imgs = ctpn_amd.weights.synthetic_images(1, h, w, 5)
    info1 = np.array([h, w, 1.0], np.float32)
    with ctpn_amd.Context(0, 1, h, w, prec) as ctx:
        ctx.load_weights(arena)
        lines, rois = ctx.detect(imgs, mode="O", want_rois=True, line_capacity=2048)
        cp, bp = ctx.get_tensor("rpn_cls_prob_reshape"), ctx.get_tensor("rpn_bbox_pred")
    ref = N.forward(imgs, weights, keep=set())
    d_cls = float(np.abs(cp[0] - ref["rpn_cls_prob_reshape"][0]).max())
    d_box = float(np.abs(bp[0] - ref["rpn_bbox_pred"][0]).max())
    assert d_cls < 1e-3 and d_box < 1e-3
    ref_rois = P.proposal_layer(ref["rpn_cls_prob_reshape"], ref["rpn_bbox_pred"], info1)
    frac = match_rois(rois[0], ref_rois, px_tol=1.0, score_tol=1e-3)
    assert frac >= 0.98
    exact = P.proposal_layer(cp[0:1], bp[0:1], info1)                  # exact given the device's own heads
    assert rois[0].shape == exact.shape and np.array_equal(rois[0][:, 0], exact[:, 0]) and np.abs(rois[0] - exact).max() < 1e-3
    assert match_lines(lines[0], P.text_detect(exact[:, 1:5], exact[:, 0], (h, w), "O"), 1.0, 1e-3)
    ref_lines = P.text_detect(ref_rois[:, 1:5], ref_rois[:, 0], (h, w), "O")
    same = bool(match_lines(lines[0], ref_lines, 1.0, 1e-3))
    print("config 5 geometry, %s: cls_prob |diff| %.2e, bbox |diff| %.2e, roi match %.4f, %d lines (oracle %d), oracle-identical lines: %s"
          % (prec, d_cls, d_box, frac, len(lines[0]), len(ref_lines), same))
    assert d_cls < 2e-4 and frac >= 0.995 and same


def test_lone_image_through_the_synchronous_seams_equals_the_batch_path(arena):
    """The reference's calling convention (ctpn/demo.py:55-68): one image per call, nothing in flight. ctpn_detect on a batch-1 ctx, the
    drop-in pair test_ctpn + TextDetector.detect, and image 3 of a batch of 4 give the same text lines, bit for bit (same kernels' sums)."""
    from ctpn_amd.lib.fast_rcnn.config import cfg
    from ctpn_amd.lib.fast_rcnn.test import test_ctpn
    from ctpn_amd.lib.networks.factory import get_network
    from ctpn_amd.lib.text_connector.detectors import TextDetector
    imgs = ctpn_amd.weights.synthetic_images(4, 600, 900, 21)
    with ctpn_amd.Context(0, 4, 600, 900, "bf16") as ctx:
        ctx.load_weights(arena)
        batch_lines, batch_rois = ctx.detect(imgs, want_rois=True)
    with ctpn_amd.Context(0, 1, 600, 900, "bf16") as ctx:
        ctx.load_weights(arena)
        for i in (3, 0, 3):
            lines, rois = ctx.detect(imgs[i:i + 1], want_rois=True)
            assert np.array_equal(lines[0], batch_lines[i]) and np.array_equal(rois[0], batch_rois[i])
    cfg.TEST.PRECISION = "bf16"
    net = get_network("VGGnet_test")
    net.load_arena(arena)
    try:
        scores, boxes = test_ctpn(None, net, imgs[3])
        assert np.array_equal(scores, batch_rois[3][:, 0]) and np.array_equal(boxes, batch_rois[3][:, 1:5])
        recs = TextDetector().detect(boxes, scores[:, np.newaxis], (600, 900))
        assert np.array_equal(recs, batch_lines[3])
    finally:
        net.close()


def test_eight_ranks_on_one_device_self_launched():
    """`python bench.py --gpus 8 --all-ranks-device 0 --batch 2`: the N = 8 code path of the driver's scaling run on a one-GPU box -- eight
    processes, eight per_rank entries, the host-thread budget divided by eight, the gloo arena path, a clean exit and ONE JSON line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--all-ranks-device", "0", "--batch", "2", "--steps", "3",
                        "--warmup", "1", "--cpu-images", "0", "--baseline-value", "1000"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["scaling"] == "weak" and out["config"]["global_batch"] == 16
    assert len(out["per_rank"]["ms_per_step"]) == 8 and len(out["per_rank"]["weight_broadcast_ms"]) == 8
    assert "gloo" in out["config"]["weight_broadcast"]
    cores = os.cpu_count() or 1
    assert 1 <= out["config"]["host_threads_per_rank"] <= 32 and out["config"]["host_threads_per_rank"] * 8 <= max(cores, 8)
    assert out["value"] > 0 and abs(out["weak_scaling_efficiency"] - out["value"] / 8000.0) < 1e-3
    assert "cpu_baseline" not in out and "other_configs" not in out          # N = 1 only


def test_small_batch_proposal_forms_sweep_against_the_generic_kernels_and_the_oracle():
    """The proposal layer from HOST heads (ctpn_proposals_from_host on a post-processing ctx) over a sweep the network path never takes:
    feature maps from 3 x 5 to 110 x 20 cells (the last one has more candidates per column than the per-column kernel's list: it must fall
    back by itself), pre / post top-N far below and above the candidate count, thresholds 0.3 .. 0.9, min sizes that filter nothing / most /
    everything, an im_info narrower than the map (boxes clipped onto one pixel column: one-workgroup form), saturated and all-equal scores
    (every key ties except for its anchor index). For one to four images per call: option nms_columns = 1 (segmented sort + one column per
    wave), 2 (one workgroup per image), 0 (generic NMS) give identical rois AND anchors, and image 0 equals the oracle's proposal_layer."""
    rng = np.random.default_rng(55)
    cases = [
        # hf, wf, pre, post, thr, min_size, (im_h, im_w) or None, score kind
        (37, 56, 12000, 1000, 0.7, 8.0, None, "random"),
        (37, 56, 300, 50, 0.7, 8.0, None, "random"),
        (37, 56, 12000, 1000, 0.3, 8.0, None, "saturated"),
        (37, 56, 12000, 1000, 0.9, 8.0, None, "equal"),
        (37, 56, 12000, 1000, 0.7, 40.0, None, "random"),       # min size above the anchors' width: every key invalid
        (37, 56, 12000, 1000, 0.7, 14.0, None, "random"),
        (37, 56, 12000, 1000, 0.7, 8.0, (592, 500), "random"),  # narrower than the map: clipped x
        (3, 5, 12000, 1000, 0.7, 8.0, None, "random"),
        (25, 80, 7000, 1000, 0.5, 8.0, None, "random"),
        (64, 50, 12000, 600, 0.7, 8.0, None, "saturated"),
        (101, 20, 12000, 1000, 0.7, 8.0, None, "random"),       # 1010 candidates per column: just inside the per-column kernel's list of 1024
        (110, 20, 12000, 1000, 0.7, 8.0, None, "random"),       # 1100: past it -- the ctx falls back to the one-workgroup form by itself
    ]
    for ci, (hf, wf, pre, post, thr, ms, im, kind) in enumerate(cases):
        for n in (1, 2, 4):
            if kind == "random":
                fg = rng.random((n, hf, wf, 10), dtype=np.float32)
            elif kind == "saturated":
                fg = np.where(rng.random((n, hf, wf, 10)) < 0.6, np.float32(1.0), rng.random((n, hf, wf, 10), dtype=np.float32)).astype(np.float32)
            else:
                fg = np.full((n, hf, wf, 10), 0.5, np.float32)
            cls = np.zeros((n, hf, wf, 20), np.float32)
            cls[..., 1::2] = fg
            cls[..., 0::2] = 1.0 - fg
            bbox = (rng.standard_normal((n, hf, wf, 40)) * 0.4).astype(np.float32)
            ih, iw = im if im else (hf * 16, wf * 16)
            info = np.array([[ih, iw, 1.0]] * n, np.float32)
            got = {}
            for opt in (1, 2, 0):
                with ctpn_amd.Context(0, 4, hf * 16, wf * 16, "fp32", postproc_only=True, options={"nms_columns": opt}) as ctx:
                    got[opt] = ctx.proposals_from_host(cls, bbox, info, pre, post, thr, ms, want_anchors=True)
            for opt in (2, 0):
                for a, b in zip(got[1][0], got[opt][0]):
                    assert a.shape == b.shape and np.array_equal(a, b), (ci, n, opt, "rois")
                for a, b in zip(got[1][1], got[opt][1]):
                    assert np.array_equal(a, b), (ci, n, opt, "anchors")
            want = P.proposal_layer(cls[:1], bbox[:1], info[0], pre_nms_topn=pre, post_nms_topn=post, nms_thresh=thr, min_size=ms)
            r0 = got[1][0][0]
            # the oracle orders ties like the device does (descending score, ascending anchor index: DESIGN section 3), so even the saturated
            # and the all-equal cases agree row for row
            assert r0.shape == want.shape and (r0.size == 0 or np.abs(r0 - want).max() < 1e-3), (ci, n, kind)
