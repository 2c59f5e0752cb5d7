"""im_list_to_blob with the reference's contract (lib/utils/blob.py:6-19): zero-padded NHWC float32 batch."""
import numpy as np


def im_list_to_blob(ims):
    shapes = np.array([im.shape for im in ims])
    hmax, wmax = int(shapes[:, 0].max()), int(shapes[:, 1].max())
    blob = np.zeros((len(ims), hmax, wmax, 3), dtype=np.float32)
    for i, im in enumerate(ims):
        blob[i, :im.shape[0], :im.shape[1], :] = im
    return blob
