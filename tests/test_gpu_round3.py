"""GPU (-m gpu), round 3: the holes VERDICT r2 listed.

  * the recurrent kernels alone, on the DEVICE's own lstm_pre, at the headline geometry (600x900, n = 16: 592 rows, T = 56) and at
    config 5's (1280x1920, n = 2: T = 120), against N.bilstm_from_pre -- round 2 only pinned them tightly at T = 6;
  * the demo path end to end on images that are NOT at network resolution -- resize_im up / down and the double resize of a
    600x1200-class image (demo.py:21-25 caps the long side at 1200, test.py:17-24 then rescales to 1000) -- against
    resize_ref -> N.forward(blob) -> P.proposal_layer -> boxes / im_scale -> P.text_detect -> draw_boxes_lines(scale);
  * the fp32 correctness gate at n = 8 (the batch the fp32 throughput line of bench.py's other_configs uses);
  * the float-blob feed against the uint8 feed in bf16 mode (they run different conv1_1 kernels);
  * the N > 1 branch of bench.py on real HIP: two ranks on device 0, gloo as the side channel;
  * the RCCL entry points of the C ABI as far as one GPU allows (world size 1);
  * a text line longer than 256 proposals through connect_kernel (recursive pairwise sum).
Nothing here reads /root/reference.
"""
import json
import os
import socket
import subprocess
import sys

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


@pytest.fixture(autouse=True)
def _default_kernel_selection():
    keys = ("CTPN_KEEP_ACTS", "CTPN_LSTM_SPLIT")
    old = {k: os.environ.get(k) for k in keys}
    yield
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,h,w", [(16, 600, 900), (2, 1280, 1920)])
def test_recurrent_kernels_on_device_pre_activations_at_benchmark_geometry(arena, weights, n, h, w):
    """bilstm_kernel (exact-fp32 MFMA; with v_exp / v_rcp gates in bf16 mode) and bilstm_split_kernel on the device's own lstm_pre.
    fp32 mode: < 5e-6 (the tolerance the toy-map test holds); fast gates / split-bf16: < 2e-5 and < 3e-5."""
    imgs = ctpn_amd.weights.synthetic_images(n, h, w, 1)
    os.environ["CTPN_KEEP_ACTS"] = "1"
    for prec, split, tol in (("fp32", "0", 5e-6), ("bf16", "0", 2e-5), ("bf16", "1", 3e-5)):
        os.environ["CTPN_LSTM_SPLIT"] = split
        with ctpn_amd.Context(0, n, h, w, prec) as ctx:
            ctx.load_weights(arena)
            ctx.forward(imgs)
            pre, out = ctx.get_tensor("lstm_pre"), ctx.get_tensor("lstm_out")
        assert pre.shape == (n, h // 16, w // 16, 1024) and out.shape == (n, h // 16, w // 16, 256)
        want = N.bilstm_from_pre(pre, weights)
        err = float(np.abs(out - want).max())
        print("recurrence %s split=%s at %dx%d n=%d: max |diff| %.2e (|out| max %.3f)" % (prec, split, h, w, n, err, float(np.abs(want).max())))
        assert np.abs(want).max() > 0.1
        assert err < tol, (prec, split, err)


# ---------------------------------------------------------------------------------------------------------------
def _oracle_demo_lines(img_bgr, weights, mode):
    """ctpn/demo.py:55-65 + lib/fast_rcnn/test.py:7-58 restated with the oracle: the text of res_<stem>.txt for one image."""
    from ctpn_amd.ctpn import demo
    from ctpn_amd.lib.fast_rcnn import test as T
    f = demo.resize_factor(img_bgr.shape, 600, 1200)                     # factors pinned to the reference's own (tests/golden/helpers.npz)
    im1 = R.resize_linear(img_bgr, f, f) if f != 1.0 else img_bgr
    s = T._scale_for(im1.shape)
    blob = im1.astype(np.float32, copy=True)
    blob -= N.PIXEL_MEANS
    blob = R.resize_linear(blob, s, s)                                   # identity when the rescale is (cv2 then returns a copy)
    out = N.forward(None, weights, keep=set(), blob=blob[None])
    info = np.array([blob.shape[0], blob.shape[1], s], np.float32)
    rois = P.proposal_layer(out["rpn_cls_prob_reshape"], out["rpn_bbox_pred"], info)
    boxes = rois[:, 1:5] / np.float32(s)
    recs = P.text_detect(boxes, rois[:, 0], im1.shape[:2], mode)
    return P.draw_boxes_lines(recs, f), im1.shape, blob.shape, s


@pytest.mark.parametrize("tag,h,w", [("up_1p25", 480, 640), ("down_portrait", 900, 700), ("double_resize", 300, 600), ("up_2p4_small", 250, 300)])
def test_demo_end_to_end_on_images_off_network_resolution(tmp_path, arena, weights, tag, h, w):
    """VERDICT r2 missing #4 / weak #3: `demo.ctpn` (imread -> resize_im -> test_ctpn -> TextDetector -> draw_boxes) in fp32 on inputs
    that need a real resize, incl. the double resize. Same lines as the oracle chain up to +-1 px and at most one borderline line."""
    pytest.importorskip("PIL")
    from PIL import Image
    from ctpn_amd.ctpn import demo
    from ctpn_amd.lib.networks.factory import get_network
    from ctpn_amd.lib.fast_rcnn.config import cfg
    bgr = ctpn_amd.weights.synthetic_images(1, h, w, 90 + h % 7)[0]
    # smooth the noise a little so that the resampled image keeps structure at every scale (pure noise averages out when shrinking)
    bgr = ((bgr.astype(np.uint16) + np.roll(bgr, 1, 0) + np.roll(bgr, 1, 1) + np.roll(bgr, (1, 1), (0, 1))) // 4).astype(np.uint8)
    path = str(tmp_path / ("%s.png" % tag))
    Image.fromarray(bgr[:, :, ::-1].copy()).save(path)
    out_dir = tmp_path / "results"
    out_dir.mkdir()
    old_prec, old_mode = cfg.TEST.PRECISION, cfg.TEST.DETECT_MODE
    net = None
    try:
        cfg.TEST.PRECISION = "fp32"
        net = get_network("VGGnet_test")
        net.load_arena(arena)
        for mode in ("H", "O"):
            cfg.TEST.DETECT_MODE = mode
            demo.ctpn(None, net, path, out_dir=str(out_dir))
            got = (out_dir / ("res_%s.txt" % tag)).read_bytes().decode()
            want, shape1, blob_shape, s = _oracle_demo_lines(bgr, weights, mode)
            if tag == "double_resize":
                assert shape1[:2] == (600, 1200) and blob_shape[:2] == (500, 1000) and abs(s - 1000.0 / 1200.0) < 1e-12
            if tag == "up_1p25":
                assert shape1[:2] == (600, 800) and s == 1.0
            got_l = [l + "\n" for l in got.split("\n")[:-1]]
            assert all(l.endswith("\r\n") for l in got_l)
            gi = sorted(tuple(int(v) for v in l.strip().split(",")) for l in got_l)
            wi = sorted(tuple(int(v) for v in l.strip().split(",")) for l in want)
            print(tag, mode, "lines device / oracle:", len(gi), len(wi))
            assert abs(len(gi) - len(wi)) <= 1
            # coordinates are int(coord / scale): +-1 px at network resolution is +-1 (+ truncation) in file units for scale >= 1
            tol = 1 + int(np.ceil(1.0 / min(1.0, demo.resize_factor(bgr.shape, 600, 1200))))
            matched = sum(1 for a in gi if any(max(abs(x - y) for x, y in zip(a, b)) <= tol for b in wi))
            assert matched >= len(gi) - 1, (tag, mode, gi[:5], wi[:5])
    finally:
        cfg.TEST.PRECISION, cfg.TEST.DETECT_MODE = old_prec, old_mode
        if net is not None:
            net.close()


# ---------------------------------------------------------------------------------------------------------------
def test_fp32_correctness_gate_at_batch_8(arena, weights):
    """BASELINE.json config 2's bar (scores 1e-3, boxes +-1 px) on a BATCH of eight 600x900 images: the configuration whose throughput
    bench.py reports under other_configs.fp32_gate_b8 (n = 8 gives every persistent workgroup several tiles; n = 1 does not)."""
    n = 8
    imgs = ctpn_amd.weights.synthetic_images(n, 600, 900, 1)
    info1 = np.array([600, 900, 1.0], np.float32)
    with ctpn_amd.Context(0, n, 600, 900, "fp32") as ctx:
        ctx.load_weights(arena)
        lines, rois = ctx.detect(imgs, want_rois=True)
        cp, bp = ctx.get_tensor("rpn_cls_prob_reshape"), ctx.get_tensor("rpn_bbox_pred")
    worst = 0.0
    for i in range(n):
        ref = N.forward(imgs[i:i + 1], weights, keep=set())
        worst = max(worst, float(np.abs(cp[i] - ref["rpn_cls_prob_reshape"][0]).max()))
        assert np.abs(cp[i] - ref["rpn_cls_prob_reshape"][0]).max() < 1e-3
        assert np.abs(bp[i] - ref["rpn_bbox_pred"][0]).max() < 1e-3
        ref_rois = P.proposal_layer(ref["rpn_cls_prob_reshape"], ref["rpn_bbox_pred"], info1)
        assert match_rois(rois[i], ref_rois, px_tol=1.0, score_tol=1e-3) >= 0.98
        exact = P.proposal_layer(cp[i:i + 1], bp[i:i + 1], info1)                  # exact given the device's own heads
        assert rois[i].shape == exact.shape and np.array_equal(rois[i][:, 0], exact[:, 0]) and np.abs(rois[i] - exact).max() < 1e-3
        assert match_lines(lines[i], P.text_detect(exact[:, 1:5], exact[:, 0], (600, 900), "H"), 1.0, 1e-3)
    print("fp32 gate n=8: worst cls_prob |diff| %.2e" % worst)


def test_float_blob_feed_tracks_uint8_feed_in_bf16_mode(arena):
    """in bf16 mode the uint8 feed runs conv1_1 as exact integer pixels x bf16-ROUNDED weights (through the q-image), the
    float32 blob feed (ctpn_forward_blob: arbitrary floats) as split-bf16, fp32-class. The two feeds of the same image therefore differ
    by conv1_1's weight rounding -- the same class of error as every other bf16 layer; bounded here (stated in include/ctpn_hip.h)."""
    imgs = ctpn_amd.weights.synthetic_images(2, 600, 900, 5)
    with ctpn_amd.Context(0, 2, 600, 900, "bf16") as ctx:
        ctx.load_weights(arena)
        info = np.array([[600, 900, 1.0]] * 2, np.float32)
        ctx.forward(imgs)
        ctx.proposals(info)                                  # the pair softmax is fused into the decode kernel
        a = ctx.get_tensor("rpn_cls_prob_reshape")
        ctx.forward_blob(N.image_blob(imgs))
        ctx.proposals(info)
        b = ctx.get_tensor("rpn_cls_prob_reshape")
    d = np.abs(a - b)
    print("bf16 uint8 feed vs float feed: cls_prob max |diff| %.3e mean %.3e" % (float(d.max()), float(d.mean())))
    assert d.max() < 3e-2 and d.mean() < 2e-3


# ---------------------------------------------------------------------------------------------------------------
def test_bench_n2_branch_runs_on_one_gpu():
    """the N > 1 code of bench.py on real HIP -- init_process_group, weight hand-over -> ctpn_load_weights, shard_range,
    per-rank timing, all_gather, barrier + MAX. Two ranks, both on device 0 (--all-ranks-device 0; RCCL refuses two ranks on one GPU, so
    the arena travels over gloo here; the RCCL entry points are covered at world size 1 below)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2", LOCAL_WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "4", "--all-ranks-device", "0",
           "--cpu-images", "0"]
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(cmd, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT))
    outs = []
    for p in procs:
        try:
            o, err = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, o, err))
    for rc, o, err in outs:
        assert rc == 0, err[-2000:]
    line = [l for l in outs[0][1].splitlines() if l.startswith("{")]
    assert len(line) == 1 and not [l for l in outs[1][1].splitlines() if l.startswith("{")]      # rank 0 alone prints the JSON line
    d = json.loads(line[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8 and d["config"]["images_per_gpu"] == 4
    assert len(d["per_rank"]["ms_per_step"]) == 2 and all(v > 0 for v in d["per_rank"]["ms_per_step"])
    assert d["value"] > 0 and abs(d["value"] - 8 * 3 / (d["ms_per_step"] * 3e-3)) < 0.02 * d["value"]
    assert "gloo" in d["config"]["weight_broadcast"]
    assert d["config"]["lines_rank0_last_step"] >= 0 and "cpu_baseline" not in d


def _run_two_ranks(extra):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2", LOCAL_WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2", "--all-ranks-device", "0",
           "--cpu-images", "0"] + extra
    procs = [subprocess.Popen(cmd, env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT)
             for r in range(2)]
    outs = []
    for p in procs:
        try:
            o, err = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, o, err))
    return outs


def test_bench_falls_back_loudly_when_the_rccl_broadcast_fails():
    """The RCCL broadcast of the C ABI attempted where it MUST fail (two ranks on one GPU: RCCL rejects duplicate devices): both ranks
    get an error (not a hang), say so on stderr, agree on the fallback through the side channel and finish the run over the gloo host
    broadcast; the JSON line records which path carried the weights."""
    outs = _run_two_ranks(["--try-rccl-on-shared-device"])
    for rc, o, err in outs:
        assert rc == 0, err[-2000:]
        assert "falling back to a host broadcast over gloo" in err
    d = json.loads([l for l in outs[0][1].splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["value"] > 0
    assert "RCCL path failed" in d["config"]["weight_broadcast"]


def test_rccl_entry_points_world_size_one(arena):
    """ctpn_comm_unique_id / ctpn_broadcast_weights_rank / ctpn_broadcast_weights through RCCL as far as one GPU allows: librccl loads
    (one copy: the one torch already mapped), a world-1 communicator forms, the broadcast runs on the ctx stream, and the ctx still
    computes the same bytes afterwards. Two ctxs on ONE device are rejected with an argument error, not a hang."""
    imgs = ctpn_amd.weights.synthetic_images(1, 96, 160, 3)
    with ctpn_amd.Context(0, 1, 96, 160, "bf16") as ctx:
        ctx.load_weights(arena)
        ctx.forward(imgs)
        before = ctx.get_tensor("heads")
        uid = B.comm_unique_id()
        assert len(uid) == B.COMM_ID_BYTES and any(uid)
        ctx.broadcast_weights_rank(uid, 0, 1, root=0)
        B.broadcast_weights([ctx])
        ctx.forward(imgs)
        assert np.array_equal(before, ctx.get_tensor("heads"))
        with ctpn_amd.Context(0, 1, 96, 160, "bf16") as other:
            with pytest.raises(ctpn_amd.CtpnError) as e:
                B.broadcast_weights([ctx, other])
            assert e.value.code == -1 and "same device" in str(e.value)
            with pytest.raises(ctpn_amd.CtpnError) as e:
                other.broadcast_weights_rank(uid, 0, 1, root=0)            # a root without weights
            assert e.value.code == -3
    maps = open("/proc/self/maps").read()
    assert len({l.split()[-1] for l in maps.splitlines() if "librccl" in l}) == 1


# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("ncols", [300, 520])
def test_device_connector_on_chains_longer_than_256_proposals(ncols):
    """numpy's pairwise float32 sum splits recursively above 128 elements; connect_kernel split only once, so the line score
    / mean height of a chain of more than 256 proposals (an image wider than 4096 px) left the host connector and the reference. One
    text line across `ncols` 16-px columns: device connector == host C++ connector == oracle, both modes."""
    rng = np.random.default_rng(ncols)
    x1 = 16.0 * np.arange(ncols, dtype=np.float32)
    y1 = (20.0 + rng.uniform(-1.0, 1.0, ncols)).astype(np.float32)
    y2 = (52.0 + rng.uniform(-1.0, 1.0, ncols)).astype(np.float32)
    scores = np.sort(rng.uniform(0.905, 0.999, ncols).astype(np.float32))[::-1].copy()
    perm = rng.permutation(ncols)                                           # score order is not column order
    boxes = np.stack([x1[perm], y1[perm], x1[perm] + 15.0, y2[perm]], 1).astype(np.float32)
    rois = np.hstack([scores[:, None], boxes]).astype(np.float32)
    size = (80, 16 * ncols)
    for mode in "HO":
        want = P.text_detect(boxes, scores, size, mode)
        host = B.text_lines(boxes, scores, size, mode, device_id=0)
        dev = B.debug_connect(rois, size, mode, scale=1.0)
        assert want.shape == (1, 9), want.shape
        assert host.shape == dev.shape == want.shape
        assert np.array_equal(dev, host), (mode, dev, host)
        assert dev[0, 8] == want[0, 8]                                      # the line score: numpy's pairwise sum, bit for bit
        assert np.allclose(dev, want, rtol=1e-6, atol=1e-3)


@pytest.mark.gpu
def test_first_bf16_forward_of_a_process_equals_the_second():
    """Round-3 regression: the tile-claim counters of conv3x3_wr were zeroed with a null-stream hipMemset, which the non-blocking
    ctx stream does not wait for -- the FIRST bf16 forward of a process that had already run the fp32 tests came out different from
    every later one. The reproducer is that exact sequence in a fresh process (the counters are allocated once per process and
    device): it failed 4 of 4 times with the old memset and never with the stream-ordered one (smaller preludes did not reproduce)."""
    sel = "fp32_every or test_full_600x900_fp32_correctness_gate or batch_equals"
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-m", "gpu", "-q", "-x", "-k", sel],
                         cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "3 passed" in out.stdout, (out.stdout[-1500:], out.stderr[-300:])      # (three tests since round 4: the im2col parametrisations of fp32_every went with CTPN_CONV_IMPL)


@pytest.mark.gpu
def test_stacked_tile_rows_batch_of_nine_equals_its_images_alone(arena):
    """From 8 images up the 16-row-patch conv kernel (conv4_1 / conv4_2 at 600x900) tiles the stacked bordered rows of the whole batch
    (tiles straddle image boundaries; border rows inside a tile are computed and not stored). One image alone never stacks, so the
    batch must reproduce its images' solo results bit for bit, twice in a row (zero borders intact after the first pass)."""
    n = 9
    imgs = ctpn_amd.weights.synthetic_images(n, 600, 900, 11)
    with ctpn_amd.Context(0, n, 600, 900, "bf16") as ctx:
        ctx.load_weights(arena)
        l1, r1 = ctx.detect(imgs, want_rois=True)
        l2, r2 = ctx.detect(imgs, want_rois=True)
        for i in range(n):
            assert np.array_equal(r1[i], r2[i]) and np.array_equal(l1[i], l2[i]), i
        for i in (0, 4, n - 1):
            ls, rs = ctx.detect(imgs[i:i + 1], want_rois=True)
            assert np.array_equal(rs[0], r1[i]) and np.array_equal(ls[0], l1[i]), i


@pytest.mark.gpu
@pytest.mark.parametrize("prec,tol", [("fp32", 5e-6), ("bf16", 8e-3)])
@pytest.mark.parametrize("shape", [(3, 17, 48), (5, 21, 112), (4, 37, 96), (9, 75, 112), (2, 16, 144), (3, 30, 225)])
def test_stacked_tile_rows_layer_equals_images_alone(prec, tol, shape):
    """One un-pooled conv layer (ctpn_debug_conv3x3) on batches whose tiles run over the stacked bordered rows of the whole batch
    (tiles straddle image boundaries; with H = 17 even an 8-row tile can): against the oracle conv, and bit for bit against the same
    images run one at a time (a single image never stacks) -- for both patch shapes and the shapes where stacking does not pay."""
    n, h, w = shape
    ci, co = 128, 128
    rng = np.random.default_rng(n * 100000 + h * 1000 + w)
    x = np.maximum(rng.standard_normal((n, h, w, ci)).astype(np.float32), 0)
    wt = (rng.standard_normal((3, 3, ci, co)) * (2.0 / (9 * ci)) ** 0.5).astype(np.float32)
    b = (rng.standard_normal(co) * 0.1).astype(np.float32)
    if prec == "bf16":
        u = x.view(np.uint32).astype(np.uint64)
        x = (((u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000).astype(np.uint32)).view(np.float32)
    full, _ = B.debug_conv3x3(x, wt, b, prec, 1, False, True)
    want = N.conv3x3_relu(x, wt, b)
    assert np.abs(full - want).max() <= tol * max(1.0, float(np.abs(want).max()))
    for i in range(n):
        alone, _ = B.debug_conv3x3(x[i:i + 1], wt, b, prec, 1, False, True)
        assert np.array_equal(alone[0], full[i]), i


