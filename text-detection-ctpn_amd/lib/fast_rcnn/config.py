"""Global `cfg` for the inference hot path; same key names, defaults and merge rules as the reference's
lib/fast_rcnn/config.py (cfg :7-226, _merge_a_into_b :256-286, cfg_from_file :288-294, cfg_from_list :296-316), so
the reference's ctpn/text.yml is readable as-is. TRAIN.* keys are accepted and stored but nothing here reads them
(training is out of scope).
"""
import os.path as osp
from ast import literal_eval

import numpy as np


class AttrDict(dict):
    """Minimal easydict.EasyDict stand-in (easydict is not installed in this image): attribute access, nested
    dicts wrapped, tuples stored as lists (easydict 1.7 behaviour the reference's type checks rely on)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    @staticmethod
    def _wrap(v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            return AttrDict(v)
        if isinstance(v, (list, tuple)):
            return [AttrDict._wrap(x) for x in v]
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, AttrDict._wrap(v))

    def __setattr__(self, k, v):
        self[k] = v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


__C = AttrDict()
cfg = __C

__C.GPU_ID = 0
__C.IS_RPN = True
__C.ANCHOR_SCALES = [16]
__C.NCLASSES = 2
__C.USE_GPU_NMS = True
__C.IS_MULTISCALE = False
__C.IS_EXTRAPOLATING = True
__C.REGION_PROPOSAL = 'RPN'
__C.NET_NAME = 'VGGnet'
__C.SUBCLS_NAME = 'voxel_exemplars'
__C.EXP_DIR = 'default'
__C.LOG_DIR = 'default'

# accepted for text.yml compatibility; unused (training is out of scope)
__C.TRAIN = AttrDict(dict(
    restore=0, max_steps=100000, SOLVER='Momentum', OHEM=False, RPN_BATCHSIZE=256, BATCH_SIZE=128, LOG_IMAGE_ITERS=100,
    DISPLAY=10, SNAPSHOT_ITERS=5000, HAS_RPN=True, LEARNING_RATE=0.001, MOMENTUM=0.9, GAMMA=0.1, STEPSIZE=50000,
    IMS_PER_BATCH=2, BBOX_NORMALIZE_TARGETS_PRECOMPUTED=False, RPN_POSITIVE_OVERLAP=0.7, PROPOSAL_METHOD='selective_search',
    BG_THRESH_LO=0.1, PRECLUDE_HARD_SAMPLES=True, BBOX_INSIDE_WEIGHTS=[1.0, 1.0, 1.0, 1.0],
    RPN_BBOX_INSIDE_WEIGHTS=[1.0, 1.0, 1.0, 1.0], RPN_POSITIVE_WEIGHT=-1.0, FG_FRACTION=0.25, WEIGHT_DECAY=0.0005))

__C.TEST = AttrDict()
__C.TEST.checkpoints_path = "checkpoints/"
__C.TEST.DETECT_MODE = "H"          # H / O
__C.TEST.SCALES = (600,)
__C.TEST.MAX_SIZE = 1000
__C.TEST.NMS = 0.3
__C.TEST.SVM = False
__C.TEST.BBOX_REG = True
__C.TEST.HAS_RPN = True
__C.TEST.PROPOSAL_METHOD = 'selective_search'
__C.TEST.RPN_NMS_THRESH = 0.7
__C.TEST.RPN_PRE_NMS_TOP_N = 12000
__C.TEST.RPN_POST_NMS_TOP_N = 1000
__C.TEST.RPN_MIN_SIZE = 8
# additions of this build (not in the reference): arithmetic of the conv stack and the batch a ctx is sized for
# PRECISION: "split" (default: (hi, lo) bf16 pairs, 3 MFMAs per product -- holds the 1e-3 / +-1 px parity bar against the fp32 path),
# "fp32" (exact-fp32 MFMA, the correctness gate), "fp16", "bf16" (16-bit MFMA throughput modes: 3.2x the rate, outside the parity bar)
__C.TEST.PRECISION = "split"
__C.TEST.MAX_BATCH = 1

__C.DEDUP_BOXES = 1. / 16.
__C.PIXEL_MEANS = np.array([[[102.9801, 115.9465, 122.7717]]])
__C.RNG_SEED = 3
__C.EPS = 1e-14
__C.ROOT_DIR = osp.abspath(osp.join(osp.dirname(__file__), '..', '..'))
__C.DATA_DIR = osp.abspath(osp.join(__C.ROOT_DIR, 'data'))


def _merge_a_into_b(a, b):
    if not isinstance(a, AttrDict):
        return
    for k, v in a.items():
        if k not in b:
            raise KeyError('{} is not a valid config key'.format(k))
        old = b[k]
        if type(old) is not type(v):
            if isinstance(old, np.ndarray):
                v = np.array(v, dtype=old.dtype)
            elif isinstance(old, float) and isinstance(v, int) and not isinstance(v, bool):
                v = float(v)  # yaml writes 1 for 1.0; the reference would raise here, this build is lenient
            else:
                raise ValueError('Type mismatch ({} vs. {}) for config key: {}'.format(type(old), type(v), k))
        if isinstance(v, AttrDict):
            try:
                _merge_a_into_b(a[k], b[k])
            except Exception:
                print('Error under config key: {}'.format(k))
                raise
        else:
            b[k] = v


def cfg_from_file(filename):
    """Load a yaml config file and merge it into the defaults."""
    import yaml
    with open(filename, 'r') as f:
        _merge_a_into_b(AttrDict(yaml.safe_load(f)), __C)


def cfg_from_list(cfg_list):
    """Set config keys from a flat [key, value, key, value ...] list (dotted keys)."""
    assert len(cfg_list) % 2 == 0
    for k, v in zip(cfg_list[0::2], cfg_list[1::2]):
        parts = k.split('.')
        d = __C
        for sub in parts[:-1]:
            assert sub in d
            d = d[sub]
        assert parts[-1] in d
        try:
            value = literal_eval(v)
        except Exception:
            value = v
        assert type(value) == type(d[parts[-1]]), \
            'type {} does not match original type {}'.format(type(value), type(d[parts[-1]]))
        d[parts[-1]] = value
