"""Host-side reference-compatible helpers (lib/fast_rcnn/bbox_transform.py:36-80). The product path runs the
same arithmetic inside the decode kernel; these exist because callers of the reference import them by name."""
import numpy as np


def bbox_transform_inv(boxes, deltas):
    """CTPN variant: only dy and dh are applied (x centre and width come from the anchor)."""
    boxes = boxes.astype(deltas.dtype, copy=False)
    w = boxes[:, 2] - boxes[:, 0] + 1.0
    h = boxes[:, 3] - boxes[:, 1] + 1.0
    cx = boxes[:, 0] + 0.5 * w
    cy = boxes[:, 1] + 0.5 * h
    pcy = deltas[:, 1::4] * h[:, None] + cy[:, None]
    ph = np.exp(deltas[:, 3::4]) * h[:, None]
    out = np.zeros(deltas.shape, dtype=deltas.dtype)
    out[:, 0::4] = cx[:, None] - 0.5 * w[:, None]
    out[:, 1::4] = pcy - 0.5 * ph
    out[:, 2::4] = cx[:, None] + 0.5 * w[:, None]
    out[:, 3::4] = pcy + 0.5 * ph
    return out


def clip_boxes(boxes, im_shape):
    for c, lim in ((0, im_shape[1]), (1, im_shape[0]), (2, im_shape[1]), (3, im_shape[0])):
        boxes[:, c::4] = np.maximum(np.minimum(boxes[:, c::4], lim - 1), 0)
    return boxes
