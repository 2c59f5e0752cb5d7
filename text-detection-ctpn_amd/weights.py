"""Flat weight arena: the reference's TF variables in their TF layouts, concatenated in a fixed order.

Names and shapes follow the variable scopes of the reference graph (lib/networks/network.py:91,146,166;
lib/networks/VGGnet_test.py:20-43), so a converter from a TF checkpoint / ctpn.pb is a
name-by-name copy. The same table is compiled into libctpn_hip.so (ctpn_weight_manifest); tests/test_abi.py
checks both agree.

`make_synthetic_arena` is the seeded random-init recipe used by the benchmark configs (there is no trained
checkpoint in the reference tree and no network here): He-scaled conv weights etc. -- the reference's own
sigma=0.01 initialisers give fg-prob == 0.500 for every anchor and no text lines.
"""
import numpy as np

_CONVS = [("conv1_1", 3, 64), ("conv1_2", 64, 64), ("conv2_1", 64, 128), ("conv2_2", 128, 128),
          ("conv3_1", 128, 256), ("conv3_2", 256, 256), ("conv3_3", 256, 256), ("conv4_1", 256, 512),
          ("conv4_2", 512, 512), ("conv4_3", 512, 512), ("conv5_1", 512, 512), ("conv5_2", 512, 512),
          ("conv5_3", 512, 512), ("rpn_conv/3x3", 512, 512)]


def _build_manifest():
    m = []
    for name, ci, co in _CONVS:
        m.append((name + "/weights", (3, 3, ci, co)))
        m.append((name + "/biases", (co,)))
    for d in ("fw", "bw"):
        m.append(("lstm_o/bidirectional_rnn/%s/lstm_cell/kernel" % d, (640, 512)))
        m.append(("lstm_o/bidirectional_rnn/%s/lstm_cell/bias" % d, (512,)))
    m += [("lstm_o/weights", (256, 512)), ("lstm_o/biases", (512,)),
          ("rpn_bbox_pred/weights", (512, 40)), ("rpn_bbox_pred/biases", (40,)),
          ("rpn_cls_score/weights", (512, 20)), ("rpn_cls_score/biases", (20,))]
    out, off = [], 0
    for name, shape in m:
        out.append((name, shape, off))
        off += int(np.prod(shape))
    return out, off


MANIFEST, WEIGHT_FLOATS = _build_manifest()
assert WEIGHT_FLOATS == 17893244


def arena_views(arena):
    """dict name -> ndarray view (TF layout) into a flat fp32 arena."""
    a = np.asarray(arena, dtype=np.float32).reshape(-1)
    assert a.size == WEIGHT_FLOATS
    return {name: a[off: off + int(np.prod(shape))].reshape(shape) for name, shape, off in MANIFEST}


def _trunc_normal(rng, shape, std):
    # TF truncated_normal: redraw beyond 2 sigma
    x = rng.standard_normal(size=shape)
    bad = np.abs(x) > 2.0
    while bad.any():
        x[bad] = rng.standard_normal(size=int(bad.sum()))
        bad = np.abs(x) > 2.0
    return (x * std).astype(np.float32)


def make_synthetic_arena(seed=0, head_cls_std=0.3, head_bbox_std=0.02, image_gain=1.0 / 64.0):
    """Seeded random-init weights. conv1_1 is additionally scaled by `image_gain`
    so that mean-subtracted uint8 pixels (|v| <= 140) do not saturate the LSTM gates downstream."""
    rng = np.random.default_rng(seed)
    arena = np.zeros((WEIGHT_FLOATS,), np.float32)
    v = arena_views(arena)
    for name, ci, co in _CONVS:
        std = np.sqrt(2.0 / (9.0 * ci))
        if name == "conv1_1":
            std *= image_gain
        v[name + "/weights"][...] = _trunc_normal(rng, (3, 3, ci, co), std)
    lim = np.sqrt(6.0 / (640 + 512))
    for d in ("fw", "bw"):
        v["lstm_o/bidirectional_rnn/%s/lstm_cell/kernel" % d][...] = rng.uniform(-lim, lim, size=(640, 512)).astype(np.float32)
    v["lstm_o/weights"][...] = _trunc_normal(rng, (256, 512), 0.1)
    v["rpn_bbox_pred/weights"][...] = _trunc_normal(rng, (512, 40), head_bbox_std)
    v["rpn_cls_score/weights"][...] = _trunc_normal(rng, (512, 20), head_cls_std)
    return arena


def synthetic_images(n, h, w, seed0=1):
    """uint8 BGR images, image i from numpy.random.default_rng(seed0 + i)."""
    return np.stack([np.random.default_rng(seed0 + i).integers(0, 256, size=(h, w, 3), dtype=np.uint8) for i in range(n)])
