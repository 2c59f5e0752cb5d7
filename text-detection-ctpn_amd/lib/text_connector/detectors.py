"""`TextDetector().detect(text_proposals, scores, size)`: the B4 seam (reference lib/text_connector/detectors.py:11-35).
Mode is read from cfg.TEST.DETECT_MODE at construction like the reference (:12-16). The whole of detect() --
score filter, sort, NMS 0.2 (on the GPU selected by cfg.GPU_ID, as nms_wrapper would), graph build, chain
extraction, line fit, filter_boxes -- is one C-ABI call, `ctpn_text_lines`."""
import numpy as np

from ..fast_rcnn.config import cfg
from ..._binding import connector_constants, text_lines
from .text_connect_cfg import Config as TextLineCfg


def check_connector_config():
    """The reference reads TextLineCfg at run time (text_proposal_graph_builder.py, text_proposal_connector*.py, detectors.py:23-31), so
    editing it changes what its detector does. Here those constants are compiled into libctpn_hip.so (host connector and device kernels):
    an edited value would silently do nothing, so it is an error instead."""
    c = TextLineCfg
    mine = {"TEXT_PROPOSALS_WIDTH * MIN_NUM_PROPOSALS": c.TEXT_PROPOSALS_WIDTH * c.MIN_NUM_PROPOSALS, "MIN_RATIO": c.MIN_RATIO,
            "LINE_MIN_SCORE": c.LINE_MIN_SCORE, "MAX_HORIZONTAL_GAP": c.MAX_HORIZONTAL_GAP, "TEXT_PROPOSALS_MIN_SCORE": c.TEXT_PROPOSALS_MIN_SCORE,
            "TEXT_PROPOSALS_NMS_THRESH": c.TEXT_PROPOSALS_NMS_THRESH, "MIN_V_OVERLAPS": c.MIN_V_OVERLAPS, "MIN_SIZE_SIM": c.MIN_SIZE_SIM}
    for name, built in connector_constants().items():
        if abs(float(mine[name]) - built) > 1e-6 * max(1.0, abs(built)):
            raise ValueError("text_connect_cfg.Config: %s = %r, but libctpn_hip.so was built with %r (the connector's constants are compiled "
                             "in: csrc/text_connector.cpp, csrc/proposal.hip)" % (name, mine[name], built))


class TextDetector:
    def __init__(self):
        self.mode = cfg.TEST.DETECT_MODE
        if self.mode not in ("H", "O"):
            raise ValueError("cfg.TEST.DETECT_MODE must be 'H' or 'O'")
        check_connector_config()

    def detect(self, text_proposals, scores, size):
        boxes = np.ascontiguousarray(text_proposals, dtype=np.float32).reshape(-1, 4)
        sc = np.ascontiguousarray(scores, dtype=np.float32).reshape(-1)
        return text_lines(boxes, sc, size, self.mode, device_id=cfg.GPU_ID if cfg.USE_GPU_NMS else -1)
