"""`nms(dets, thresh)`: same signature and return convention as the reference dispatcher
(lib/fast_rcnn/nms_wrapper.py:11-20): indices into the caller's dets, descending score; [] for empty input.
There is exactly one implementation here, the HIP kernel; a missing library or GPU raises (no py_cpu_nms fallback).
"""
from .config import cfg
from ..utils.gpu_nms import gpu_nms


def nms(dets, thresh):
    if dets.shape[0] == 0:
        return []
    return gpu_nms(dets, thresh, device_id=cfg.GPU_ID)
