"""The 10 fixed-width CTPN anchors (reference lib/rpn_msr/generate_anchors.py:3-32): width 16, heights
11..283, centre 7.5, int32 truncation toward zero, python-3 true division. The same table is
baked into the decode kernel (csrc/proposal.hip: c_anchor_y1 / c_anchor_y2); tests check both against the
fixture generated from the reference."""
import numpy as np

_HEIGHTS = (11, 16, 23, 33, 48, 68, 97, 139, 198, 283)


def generate_anchors(base_size=16, ratios=None, scales=None):
    """`ratios` and `scales` are accepted and ignored, as in the reference."""
    ctr = (base_size - 1) * 0.5
    rows = [[ctr - 16 / 2, ctr - h / 2, ctr + 16 / 2, ctr + h / 2] for h in _HEIGHTS]
    return np.trunc(np.array(rows)).astype(np.int32)
