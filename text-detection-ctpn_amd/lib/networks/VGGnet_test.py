"""The inference network object the demo passes around as `net` (reference lib/networks/VGGnet_test.py:6-55).

In the reference this class builds a TF1 graph; here it owns a `Context` (one GPU, one stream, HBM arena) whose
forward is the hand-written HIP pipeline:
    conv1_1 .. conv5_3 (+4 max-pools) -> rpn_conv/3x3 -> Bilstm(512,128,512) -> lstm_fc 512->40 / 512->20
    -> pairwise softmax -> proposal layer            (VGGnet_test.py:20-55)
Weights enter through `load(path)` (a flat fp32 arena, see ctpn_amd.weights) or `load_arena(array)`;
`restore_synthetic(seed)` gives the seeded random-init weights used by tests and benchmarks.
"""
import numpy as np

from ..fast_rcnn.config import cfg
from ..._binding import Context
from ... import weights as _w


class VGGnet_test(object):
    def __init__(self, trainable=False, max_batch=None, max_h=None, max_w=None, precision=None, device_id=None):
        self.trainable = trainable
        self.max_batch = max_batch or int(cfg.TEST.MAX_BATCH)
        # largest blob _get_image_blob can emit: short side SCALES[0], long side capped at MAX_SIZE (test.py:16-24)
        self.max_h = max_h or int(max(cfg.TEST.MAX_SIZE, cfg.TEST.SCALES[0]))
        self.max_w = max_w or int(max(cfg.TEST.MAX_SIZE, cfg.TEST.SCALES[0]))
        self.precision = precision or cfg.TEST.PRECISION
        self.device_id = cfg.GPU_ID if device_id is None else device_id
        self._ctx = None
        self._arena = None

    # -- lifecycle -----------------------------------------------------------------------------
    @property
    def ctx(self):
        if self._ctx is None:
            self._ctx = Context(self.device_id, self.max_batch, self.max_h, self.max_w, self.precision)
            if self._arena is not None:
                self._ctx.load_weights(self._arena)
        return self._ctx

    def ensure_capacity(self, n, h, w):
        if n > self.max_batch or h > self.max_h or w > self.max_w:
            self.max_batch, self.max_h, self.max_w = max(n, self.max_batch), max(h, self.max_h), max(w, self.max_w)
            if self._ctx is not None:
                self._ctx.close()
                self._ctx = None

    def close(self):
        if self._ctx is not None:
            self._ctx.close()
            self._ctx = None

    # -- weights -------------------------------------------------------------------------------
    def load_arena(self, arena):
        self._arena = np.ascontiguousarray(arena, dtype=np.float32).reshape(-1)
        if self._ctx is not None:
            self._ctx.load_weights(self._arena)
        return self

    def load(self, data_path, session=None, ignore_missing=False):
        """Reference name (Network.load, lib/networks/network.py:40-53). Accepts a frozen graph (.pb, ctpn/demo_pb.py:60-66), a
        .npy flat arena, an .npz / .npy dict keyed by the TF variable names, or the nested VGG_imagenet.npy layout
        (weights_import.py)."""
        from ... import weights_import as _wi
        if str(data_path).endswith(".pb") or not ignore_missing:
            return self.load_arena(_wi.load_any(str(data_path), base=self._arena))
        arena, _ = _wi.arena_from_vgg_npy(str(data_path), base=self._arena)     # ignore_missing: the VGG_imagenet.npy initialisation
        return self.load_arena(arena)

    def restore_synthetic(self, seed=0):
        return self.load_arena(_w.make_synthetic_arena(seed))

    @property
    def has_weights(self):
        return self._arena is not None
