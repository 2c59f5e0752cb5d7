"""`gpu_nms(dets, thresh, device_id=0)`: signature of the reference's Cython wrapper (lib/utils/gpu_nms.pyx:16-31).
Host side: order by descending score (ties by ascending index -- the reference leaves ties to numpy's unstable
argsort), gather, call the C-ABI `ctpn_nms` (B1 seam, replaces `_nms`, lib/utils/gpu_nms.hpp:1-2),
map kept positions back through the order.
"""
import numpy as np

from ..._binding import nms_sorted


def gpu_nms(dets, thresh, device_id=0):
    dets = np.ascontiguousarray(dets, dtype=np.float32)
    if dets.shape[0] == 0:
        return []
    scores = dets[:, 4]
    order = np.lexsort((np.arange(scores.size), -scores.astype(np.float64)))
    keep = nms_sorted(dets[order, :], float(thresh), int(device_id))
    return list(order[keep])
