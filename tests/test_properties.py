"""CPU property / differential tests (hypothesis), as SURVEY.md section 8c suggests for the two combinatorial parts of the path:
greedy NMS (reference lib/fast_rcnn/nms_wrapper.py:23-47, lib/utils/cython_nms.pyx:17-68) and the text connector
(reference lib/text_connector/*). Generated inputs are CTPN-shaped (16 px wide boxes on a 16 px grid, fp32 scores with ties).
No GPU: the numpy oracle, its C twin and the product's host C++ connector (ctpn_text_lines with device_id = -1)."""
import numpy as np
from hypothesis import given, settings, strategies as st

import ctpn_amd  # noqa: F401
from ctpn_amd import _binding as B
from oracle import cnms
from oracle import postproc as P


@st.composite
def ctpn_dets(draw, max_n=120):
    """(n, 5) float32 rows [x1, y1, x2, y2, score]: anchors' x geometry, free heights, scores from a small set (ties on purpose)."""
    n = draw(st.integers(min_value=0, max_value=max_n))
    seed = draw(st.integers(min_value=0, max_value=2 ** 31 - 1))
    rng = np.random.default_rng(seed)
    col = rng.integers(0, 12, n)
    y1 = rng.integers(0, 60, n).astype(np.float32) + rng.choice([0.0, 0.25, 0.5], n).astype(np.float32)
    h = rng.integers(4, 60, n).astype(np.float32)
    score = rng.choice(np.linspace(0.05, 1.0, 12).astype(np.float32), n)
    return np.stack([16.0 * col, y1, 16.0 * col + 15.0, y1 + h, score], axis=1).astype(np.float32)


def iou(a, b):
    xx1, yy1, xx2, yy2 = max(a[0], b[0]), max(a[1], b[1]), min(a[2], b[2]), min(a[3], b[3])
    w, h = max(np.float32(0), xx2 - xx1 + 1), max(np.float32(0), yy2 - yy1 + 1)
    inter = np.float32(w * h)
    aa = np.float32((a[2] - a[0] + 1) * (a[3] - a[1] + 1)); ab = np.float32((b[2] - b[0] + 1) * (b[3] - b[1] + 1))
    return np.float32(inter / (aa + ab - inter))


@settings(max_examples=60, deadline=None)
@given(ctpn_dets(), st.sampled_from([0.2, 0.5, 0.7]))
def test_greedy_nms_properties(dets, thr):
    keep = P.nms(dets, thr)
    assert keep == cnms.nms(dets, thr, 1)                                     # numpy restatement == C restatement of py_cpu_nms
    n = len(dets)
    assert len(set(keep)) == len(keep) and all(0 <= k < n for k in keep)
    if n == 0:
        assert keep == []
        return
    sc = dets[:, 4]
    assert all(sc[keep[i]] >= sc[keep[i + 1]] for i in range(len(keep) - 1))   # descending score
    t = np.float32(thr)
    for i, a in enumerate(keep):                                               # kept boxes do not suppress each other
        for b in keep[i + 1:]:
            assert not iou(dets[a], dets[b]) > t
    kept = set(keep)
    order = P.desc_order(sc)
    rank = {int(k): r for r, k in enumerate(order)}
    for j in range(n):                                                         # every dropped box has an earlier kept suppressor
        if j not in kept:
            assert any(rank[k] < rank[j] and iou(dets[k], dets[j]) > t for k in keep), j
    again = P.nms(dets[keep], thr)                                             # idempotent on its own output
    assert again == list(range(len(keep)))


@settings(max_examples=40, deadline=None)
@given(ctpn_dets(max_n=200), st.sampled_from(["H", "O"]))
def test_host_connector_equals_oracle_on_generated_proposals(dets, mode):
    """csrc/text_connector.cpp (score filter, sort, NMS 0.2, graph, chains, line fit, filter_boxes) against oracle/postproc.py::
    text_detect, itself pinned against the reference's TextDetector on the fixtures: identical float64 records."""
    boxes, scores = dets[:, :4].copy(), dets[:, 4].copy()
    scores = np.where(scores > 0.5, np.float32(0.7) + (scores - np.float32(0.5)) * np.float32(0.59), scores).astype(np.float32)   # most above 0.7
    size = (140, 200)
    want = P.text_detect(boxes.copy(), scores[:, None].copy(), size, mode)
    got = B.text_lines(boxes, scores, size, mode, device_id=-1)
    assert got.shape == want.shape
    # The fitted coordinates may differ in the last fp32 bit: the reference's np.polyfit on float32 data is LAPACK's float32
    # least squares (whatever BLAS numpy was built with), the C++ is a double closed form rounded to fp32. Every reference
    # fixture agrees bit for bit (test_cpp_connector_matches_reference_lines); generated chains show the occasional 1-ulp case.
    assert np.array_equal(got[:, 8], want[:, 8])                                        # scores: plain fp32 means
    assert np.allclose(got[:, :8], want[:, :8], rtol=3e-7, atol=1e-5), np.abs(got - want).max()


@settings(max_examples=80, deadline=None)
@given(st.integers(min_value=0, max_value=2 ** 31 - 1), st.integers(min_value=0, max_value=40), st.sampled_from([1.0, 0.75, 1.5, 2.0, 0.6667, 1.8018018]))
def test_result_writer_equals_oracle_on_generated_records(seed, m, scale):
    """ctpn_result_text (host C++) against oracle/postproc.py::draw_boxes_lines (pinned on the reference's own draw_boxes bytes):
    the scalar skip test of demo.py:32, int() truncation towards zero (negative coordinates of oriented boxes included), '\\r\\n'."""
    rng = np.random.default_rng(seed)
    recs = rng.uniform(-40, 1300, (m, 9))
    recs[:, 8] = rng.uniform(0.9, 1.0, m)
    pick = rng.random(m) < 0.3                       # some rows trip the skip test |x1 - y1| < 5 or |y2 - x1| < 5
    recs[pick, 1] = recs[pick, 0] + rng.uniform(-6, 6, int(pick.sum()))
    want = "".join(P.draw_boxes_lines(recs, scale)).encode()
    assert B.result_text(recs, scale) == want


@settings(max_examples=200, deadline=None)
@given(st.integers(min_value=1, max_value=4000), st.integers(min_value=1, max_value=4000),
       st.floats(min_value=0.05, max_value=8.0, allow_nan=False), st.floats(min_value=0.05, max_value=8.0, allow_nan=False))
def test_resize_dims_equals_oracle(h, w, fx, fy):
    """ctpn_resize_dims (host arithmetic of cv2.resize's dsize: cvRound(src * f), half to even) against oracle/resize_ref.py."""
    from oracle import resize_ref as R
    oh, ow = R.out_dim(h, fy), R.out_dim(w, fx)
    if oh < 1 or ow < 1:
        return
    assert B.resize_dims(h, w, fx, fy) == (oh, ow)
