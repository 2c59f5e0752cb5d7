"""`proposal_layer(...)` with the reference's signature and return value (lib/rpn_msr/proposal_layer_tf.py:14-157):
    blob (R,5) float32 [score, x1, y1, x2, y2], bbox_deltas (R,4)
computed by the HIP decode -> sort -> NMS pipeline (C-ABI ctpn_proposals_from_host, the ctpn/demo_pb.py:91-92 seam).
Unlike the reference (assert batch == 1, :51-52) a leading batch > 1 is accepted and returns lists.
"""
import numpy as np

from ..fast_rcnn.config import cfg
from ..._binding import Context

_ctx_cache = {}


def _ctx_for(n, hf, wf):
    key = (cfg.GPU_ID,)
    ctx = _ctx_cache.get(key)
    need_h, need_w = hf * 16, wf * 16
    if ctx is None or ctx.max_batch < n or ctx.max_h < need_h or ctx.max_w < need_w:
        if ctx is not None:
            ctx.close()
        # post-processing only (ctpn_create_postproc): proposal-layer buffers, no VGG activation arena, no weights
        ctx = Context(cfg.GPU_ID, max(n, 1), max(need_h, 16), max(need_w, 16), postproc_only=True)
        _ctx_cache[key] = ctx
    return ctx


def proposal_layer(rpn_cls_prob_reshape, rpn_bbox_pred, im_info, cfg_key, _feat_stride=[16, ], anchor_scales=[16, ]):
    if isinstance(cfg_key, bytes):
        cfg_key = cfg_key.decode('ascii')
    cls = np.ascontiguousarray(rpn_cls_prob_reshape, dtype=np.float32)
    box = np.ascontiguousarray(rpn_bbox_pred, dtype=np.float32)
    n, hf, wf, _ = cls.shape
    c = cfg[cfg_key]
    info = np.ascontiguousarray(im_info, dtype=np.float32).reshape(-1, 3)
    ctx = _ctx_for(n, hf, wf)
    rois, anchors = ctx.proposals_from_host(cls, box, info, c.RPN_PRE_NMS_TOP_N, c.RPN_POST_NMS_TOP_N, c.RPN_NMS_THRESH,
                                            c.RPN_MIN_SIZE, want_anchors=True)
    # bbox_deltas[order][keep] (reference :133-157): the kept anchors' rows of rpn_bbox_pred.reshape(-1, 4) -- exact, the
    # anchor index of every roi comes back from the device (ctpn_proposal_anchors)
    d4 = box.reshape(n, -1, 4)
    deltas = [d4[i][a] for i, a in enumerate(anchors)]
    if n == 1:
        return rois[0], deltas[0]
    return rois, deltas
