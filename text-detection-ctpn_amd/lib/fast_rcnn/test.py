"""`test_ctpn(sess, net, im)`: the B3 seam with the reference's signature and return (lib/fast_rcnn/test.py:40-58):
    scores (R,), boxes (R,4) = rois[:,1:5] / im_scale
`sess` is accepted and ignored (there is no TF session); `net` is a lib.networks.VGGnet_test.

Pre-processing follows _get_image_blob (test.py:7-31): float32 - PIXEL_MEANS, then the TEST.SCALES / MAX_SIZE
rescale with bilinear interpolation. When that rescale is the identity (the usual case after demo.resize_im) the
uint8 image goes to the GPU as-is and the mean subtraction happens inside the conv1_1 kernel; otherwise the
float blob is built on the host like the reference does and fed through ctpn_forward_blob.
`test_ctpn_batch` is this build's batched form (list of same-size images).
"""
import numpy as np

from .config import cfg
from ..utils.image import resize_bilinear


def _scale_for(shape):
    size_min, size_max = min(shape[0:2]), max(shape[0:2])
    target = cfg.TEST.SCALES[0]
    s = float(target) / float(size_min)
    if np.round(s * size_max) > cfg.TEST.MAX_SIZE:
        s = float(cfg.TEST.MAX_SIZE) / float(size_max)
    return s


def _get_image_blob(im):
    """-> (blob (1,H,W,3) float32, im_scales (1,)); host restatement used only when a rescale is needed."""
    im_orig = im.astype(np.float32, copy=True)
    im_orig -= cfg.PIXEL_MEANS
    s = _scale_for(im_orig.shape)
    out = resize_bilinear(im_orig, fx=s, fy=s)
    return out[None], np.array([s])


def test_ctpn(sess, net, im, boxes=None):
    scores, bxs = test_ctpn_batch(net, [im])
    return scores[0], bxs[0]


def test_ctpn_batch(net, ims):
    ims = [np.asarray(im) for im in ims]
    shape0 = ims[0].shape
    assert all(im.shape == shape0 for im in ims), "a batch must hold same-size images"
    s = _scale_for(shape0)
    identity = (int(round(shape0[0] * s)) == shape0[0] and int(round(shape0[1] * s)) == shape0[1])
    c = cfg.TEST
    if identity and ims[0].dtype == np.uint8:
        batch = np.stack(ims)
        n, h, w, _ = batch.shape
        net.ensure_capacity(n, h, w)
        net.ctx.forward(batch)
        scales = np.full((n,), s if not identity else 1.0, np.float32)
    else:
        blobs = [_get_image_blob(im)[0][0] for im in ims]
        batch = np.stack(blobs).astype(np.float32)
        n, h, w, _ = batch.shape
        net.ensure_capacity(n, h, w)
        net.ctx.forward_blob(batch)
        scales = np.full((n,), s, np.float32)
    im_info = np.stack([[h, w, sc] for sc in scales]).astype(np.float32)
    rois = net.ctx.proposals(im_info, c.RPN_PRE_NMS_TOP_N, c.RPN_POST_NMS_TOP_N, c.RPN_NMS_THRESH, c.RPN_MIN_SIZE)
    return [r[:, 0] for r in rois], [r[:, 1:5] / sc for r, sc in zip(rois, scales)]
