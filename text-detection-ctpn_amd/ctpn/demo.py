"""`python ctpn/demo.py` -- same entry point, inputs and outputs as the reference's ctpn/demo.py:71-105:

    data/demo/*.png|*.jpg  ->  data/results/res_<stem>.txt  ("x1,y1,x2,y2\\r\\n" per line)  +  annotated image

run from the package root (text-detection-ctpn_amd/) like the reference is run from its repo root, or from anywhere
with --root. Weights: cfg.TEST.checkpoints_path/ctpn_weights.npy (flat fp32 arena, see ctpn_amd.weights) or
ctpn_weights.npz (dict of TF variable names); `--synthetic SEED` uses the seeded random-init weights instead
(there is no trained checkpoint in the reference tree).
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
from ctpn_amd.lib.networks.factory import get_network  # noqa: E402
from ctpn_amd.lib.fast_rcnn.config import cfg, cfg_from_file  # noqa: E402
from ctpn_amd.lib.fast_rcnn.test import test_ctpn  # noqa: E402
from ctpn_amd.lib.utils.timer import Timer  # noqa: E402
from ctpn_amd.lib.utils import image as imutil  # noqa: E402
from ctpn_amd.lib.text_connector.detectors import TextDetector  # noqa: E402
from ctpn_amd.lib.text_connector.text_connect_cfg import Config as TextLineCfg  # noqa: E402


def resize_factor(shape, scale, max_scale=None):
    """Short side -> scale, long side capped at max_scale (reference demo.py:22-24)."""
    f = float(scale) / min(shape[0], shape[1])
    if max_scale is not None and f * max(shape[0], shape[1]) > max_scale:
        f = float(max_scale) / max(shape[0], shape[1])
    return f


def resize_im(im, scale, max_scale=None):
    """reference demo.py:21-25; the cv2.resize runs on the GPU (lib/utils/image.py)."""
    f = resize_factor(im.shape, scale, max_scale)
    return imutil.resize_bilinear(im, fx=f, fy=f), f


def result_lines(boxes, scale):
    """Text of res_<stem>.txt for (M,9) records; keeps the reference's skip test on SCALARS box[0]-box[1] and
    box[3]-box[0] (demo.py:32, SURVEY.md A.5 iv) and its int() truncation (demo.py:43-49)."""
    out = []
    for box in boxes:
        if abs(box[0] - box[1]) < 5 or abs(box[3] - box[0]) < 5:
            continue
        xs = [int(box[i] / scale) for i in (0, 2, 4, 6)]
        ys = [int(box[i] / scale) for i in (1, 3, 5, 7)]
        out.append(','.join(str(v) for v in (min(xs), min(ys), max(xs), max(ys))) + '\r\n')
    return out


def draw_boxes(img, image_name, boxes, scale, out_dir='data/results'):
    """reference demo.py:28-52. The res_<stem>.txt writer and the outline rasteriser are host C++ behind the C ABI
    (ctpn_write_result_file, ctpn_draw_boxes); result_lines above is their pure-Python statement, kept for the tests."""
    from ctpn_amd import _binding as B
    base_name = image_name.split('/')[-1]
    B.write_result_file(os.path.join(out_dir, 'res_{}.txt'.format(base_name.split('.')[0])), boxes, scale)
    img = np.ascontiguousarray(img, dtype=np.uint8)
    B.draw_boxes(img, boxes)
    img = imutil.resize_bilinear(img, fx=1.0 / scale, fy=1.0 / scale)
    imutil.imwrite(os.path.join(out_dir, base_name), img)


def ctpn(sess, net, image_name, out_dir='data/results'):
    timer = Timer()
    timer.tic()
    img = imutil.imread(image_name)
    img, scale = resize_im(img, scale=TextLineCfg.SCALE, max_scale=TextLineCfg.MAX_SCALE)
    scores, boxes = test_ctpn(sess, net, img)
    boxes = TextDetector().detect(boxes, scores[:, np.newaxis], img.shape[:2])
    draw_boxes(img, image_name, boxes, scale, out_dir)
    timer.toc()
    print(('Detection took {:.3f}s for {:d} object proposals').format(timer.total_time, boxes.shape[0]))
    return boxes


def load_weights(net, synthetic_seed=None):
    if synthetic_seed is not None:
        print('Using seeded random-init weights (seed {:d})'.format(synthetic_seed))
        return net.restore_synthetic(synthetic_seed)
    # tf.train.get_checkpoint_state(cfg.TEST.checkpoints_path) (reference demo.py:84-92): the `checkpoint` text file names the
    # newest Saver-V2 prefix; weights_import reads the bundle without TensorFlow
    state = os.path.join(cfg.TEST.checkpoints_path, 'checkpoint')
    if os.path.exists(state):
        import re
        m = re.search(r'model_checkpoint_path:\s*"([^"]+)"', open(state).read())
        if m:
            prefix = m.group(1) if os.path.isabs(m.group(1)) else os.path.join(cfg.TEST.checkpoints_path, os.path.basename(m.group(1)))
            if os.path.exists(prefix + '.index'):
                print('Restoring from {}...'.format(prefix), end=' ')
                net.load(prefix)
                print('done')
                return net
    for name in ('ctpn.pb', 'ctpn_weights.npy', 'ctpn_weights.npz'):
        path = os.path.join(cfg.TEST.checkpoints_path, name)
        if os.path.exists(path):
            print('Restoring from {}...'.format(path), end=' ')
            net.load(path)
            print('done')
            return net
    raise IOError('Check your pretrained weights: no ctpn.pb / ctpn_weights.npy / .npz under {:s}'.format(cfg.TEST.checkpoints_path))


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument('--root', default=os.getcwd(), help='directory holding ctpn/text.yml, data/demo, checkpoints/')
    ap.add_argument('--synthetic', type=int, default=None, metavar='SEED')
    args = ap.parse_args(argv)
    os.chdir(args.root)
    if os.path.exists("data/results/"):
        shutil.rmtree("data/results/")
    os.makedirs("data/results/")
    yml = 'ctpn/text.yml' if os.path.exists('ctpn/text.yml') else os.path.join(os.path.dirname(os.path.abspath(__file__)), 'text.yml')
    cfg_from_file(yml)
    cfg.DATA_DIR = os.path.join(args.root, 'data')

    sess = None  # no TF session on this path
    print(('Loading network {:s}... '.format("VGGnet_test")), end=' ')
    net = get_network("VGGnet_test")
    load_weights(net, args.synthetic)

    im = 128 * np.ones((300, 300, 3), dtype=np.uint8)
    for _ in range(2):
        test_ctpn(sess, net, im)

    im_names = glob.glob(os.path.join(cfg.DATA_DIR, 'demo', '*.png')) + glob.glob(os.path.join(cfg.DATA_DIR, 'demo', '*.jpg'))
    for im_name in im_names:
        print('~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~')
        print(('Demo for {:s}'.format(im_name)))
        ctpn(sess, net, im_name)


if __name__ == '__main__':
    main()
