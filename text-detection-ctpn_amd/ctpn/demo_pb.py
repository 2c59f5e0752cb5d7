"""`python ctpn/demo_pb.py` -- the frozen-graph entry point of the reference (ctpn/demo_pb.py:55-98) with the same inputs and outputs:

    data/ctpn.pb + data/demo/*.png|*.jpg  ->  data/results/res_<stem>.txt + annotated image

The reference imports the GraphDef into a tf.Session, fetches `Reshape_2:0` (cls_prob) and `rpn_bbox_pred/Reshape_1:0` (box_pred)
and hands them to the Python proposal_layer (:91-92). Here the frozen graph is read without TensorFlow (weights_import.py: every
variable is a Const node under its scope name), the network runs on the HIP path, and the SAME seam is kept: the two head tensors
come back to the host and go through `proposal_layer(cls_prob, box_pred, im_info, 'TEST', anchor_scales=cfg.ANCHOR_SCALES)`
(lib/rpn_msr/proposal_layer_tf.py over ctpn_proposals_from_host), then TextDetector and draw_boxes exactly as in demo.py.
`--synthetic SEED` writes a frozen graph of the seeded random-init weights first (there is no trained ctpn.pb in the reference tree).
"""
from __future__ import print_function

import argparse
import glob
import os
import shutil
import sys

import numpy as np

_PKG_PARENT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _PKG_PARENT not in sys.path:
    sys.path.insert(0, _PKG_PARENT)

import ctpn_amd  # noqa: E402,F401
from ctpn_amd import weights_import as WI  # noqa: E402
from ctpn_amd.ctpn.demo import draw_boxes, resize_im  # noqa: E402
from ctpn_amd.lib.networks.factory import get_network  # noqa: E402
from ctpn_amd.lib.fast_rcnn.config import cfg, cfg_from_file  # noqa: E402
from ctpn_amd.lib.fast_rcnn.test import _scale_for, _get_image_blob  # noqa: E402
from ctpn_amd.lib.rpn_msr.proposal_layer_tf import proposal_layer  # noqa: E402
from ctpn_amd.lib.text_connector.detectors import TextDetector  # noqa: E402
from ctpn_amd.lib.text_connector.text_connect_cfg import Config as TextLineCfg  # noqa: E402
from ctpn_amd.lib.utils import image as imutil  # noqa: E402


def run_heads(net, img):
    """`sess.run([output_cls_prob, output_box_pred], feed_dict={input_img: blobs['data']})` of the reference (:91):
    -> cls_prob (1,Hf,Wf,20), box_pred (1,Hf,Wf,40), im_info (1,3), im_scale."""
    s = _scale_for(img.shape)
    identity = int(round(img.shape[0] * s)) == img.shape[0] and int(round(img.shape[1] * s)) == img.shape[1]
    if identity and img.dtype == np.uint8:
        h, w = img.shape[:2]
        net.ensure_capacity(1, h, w)
        net.ctx.forward(img[None])
        s = 1.0
    else:
        blob, scales = _get_image_blob(img)
        h, w = blob.shape[1:3]
        net.ensure_capacity(1, h, w)
        net.ctx.forward_blob(blob)
        s = float(scales[0])
    im_info = np.array([[h, w, s]], dtype=np.float32)
    # the pairwise softmax is fused into the decode kernel: one proposals call materialises both head tensors
    net.ctx.proposals(im_info, cfg.TEST.RPN_PRE_NMS_TOP_N, cfg.TEST.RPN_POST_NMS_TOP_N, cfg.TEST.RPN_NMS_THRESH, cfg.TEST.RPN_MIN_SIZE)
    return net.ctx.get_tensor("rpn_cls_prob_reshape"), net.ctx.get_tensor("rpn_bbox_pred"), im_info, s


def ctpn_pb(net, im_name, out_dir='data/results'):
    img = imutil.imread(im_name)
    img, scale = resize_im(img, scale=TextLineCfg.SCALE, max_scale=TextLineCfg.MAX_SCALE)
    cls_prob, box_pred, im_info, im_scale = run_heads(net, img)
    rois, _ = proposal_layer(cls_prob, box_pred, im_info, 'TEST', anchor_scales=cfg.ANCHOR_SCALES)
    scores = rois[:, 0]
    boxes = rois[:, 1:5] / im_scale
    boxes = TextDetector().detect(boxes, scores[:, np.newaxis], img.shape[:2])
    draw_boxes(img, im_name, boxes, scale, out_dir)
    return boxes


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('--root', default=os.getcwd(), help='directory holding ctpn/text.yml, data/ctpn.pb, data/demo')
    ap.add_argument('--synthetic', type=int, default=None, metavar='SEED', help='write data/ctpn.pb from the seeded random-init weights first')
    args = ap.parse_args(argv)
    os.chdir(args.root)
    if os.path.exists("data/results/"):
        shutil.rmtree("data/results/")
    os.makedirs("data/results/")
    yml = 'ctpn/text.yml' if os.path.exists('ctpn/text.yml') else os.path.join(os.path.dirname(os.path.abspath(__file__)), 'text.yml')
    cfg_from_file(yml)
    cfg.DATA_DIR = os.path.join(args.root, 'data')
    pb = os.path.join('data', 'ctpn.pb')
    if args.synthetic is not None:
        arena = ctpn_amd.make_synthetic_arena(args.synthetic)
        WI.write_frozen_graph(pb, dict(ctpn_amd.arena_views(arena)))
    if not os.path.exists(pb):
        raise IOError('no frozen graph at {:s} (ctpn/generate_pb.py of the reference writes it)'.format(pb))
    net = get_network("VGGnet_test")
    net.load(pb)
    im_names = glob.glob(os.path.join(cfg.DATA_DIR, 'demo', '*.png')) + glob.glob(os.path.join(cfg.DATA_DIR, 'demo', '*.jpg'))
    for im_name in im_names:
        print('~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~')
        print(('Demo for {:s}'.format(im_name)))
        ctpn_pb(net, im_name)
    net.close()


if __name__ == '__main__':
    main()
