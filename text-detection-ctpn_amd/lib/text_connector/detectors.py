"""`TextDetector().detect(text_proposals, scores, size)`: the B4 seam (reference lib/text_connector/detectors.py:11-35).
Mode is read from cfg.TEST.DETECT_MODE at construction like the reference (:12-16). The whole of detect() --
score filter, sort, NMS 0.2 (on the GPU selected by cfg.GPU_ID, as nms_wrapper would), graph build, chain
extraction, line fit, filter_boxes -- is one C-ABI call, `ctpn_text_lines`."""
import numpy as np

from ..fast_rcnn.config import cfg
from ..._binding import text_lines


class TextDetector:
    def __init__(self):
        self.mode = cfg.TEST.DETECT_MODE
        if self.mode not in ("H", "O"):
            raise ValueError("cfg.TEST.DETECT_MODE must be 'H' or 'O'")

    def detect(self, text_proposals, scores, size):
        boxes = np.ascontiguousarray(text_proposals, dtype=np.float32).reshape(-1, 4)
        sc = np.ascontiguousarray(scores, dtype=np.float32).reshape(-1)
        return text_lines(boxes, sc, size, self.mode, device_id=cfg.GPU_ID if cfg.USE_GPU_NMS else -1)
