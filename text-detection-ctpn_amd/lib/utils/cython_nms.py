"""Name kept for drop-in imports (`from lib.utils.cython_nms import nms`, reference lib/utils/cython_nms.pyx:17).
This build has no CPU NMS on the product path; the call is served by the HIP kernel."""
from ..fast_rcnn.config import cfg
from .gpu_nms import gpu_nms


def nms(dets, thresh):
    return gpu_nms(dets, thresh, device_id=cfg.GPU_ID)
