"""Generates tests/golden/jpeg_cases.npz: a dozen small JPEG files (bytes) and Pillow's decode of each in cv2.imread's channel order.

    python -m oracle.make_jpeg_golden          (from the repo root, in the build container)

The decoder behind the reference's cv2.imread (ctpn/demo.py:59) is libjpeg(-turbo); cv2 is not installed here, Pillow is and links the same
decoder family (PIL.features.version("jpg") is recorded in the file). The committed vectors pin oracle/jpeg_ref.py, the library's host half
and the device kernels to THIS decoder's output independently of the Pillow that happens to be installed where the tests run.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import PIL
    from PIL import features
    from util_jpeg import CASES, case_id, cv2_like_bgr, encode, extra_cases, pillow_bgr, scene
    out = {}
    names = []
    for c in CASES:
        h, w, q, sub, gray, kw = c
        data = encode(scene(h, w, h + w, gray), q, sub, **kw)
        cid = case_id(c)
        names.append(cid)
        out["file_" + cid] = np.frombuffer(data, np.uint8)
        out["bgr_" + cid] = pillow_bgr(data)
    # layouts Pillow cannot write (4:4:0, by tests/util_jpeg.py's own small encoder) and EXIF orientations: Pillow's decode, turned by
    # ImageOps.exif_transpose the way cv2.imread turns it
    for cid, data in extra_cases().items():
        names.append(cid)
        out["file_" + cid] = np.frombuffer(data, np.uint8)
        out["bgr_" + cid] = cv2_like_bgr(data)
    out["names"] = np.array(names)
    out["decoder"] = np.array("Pillow %s, libjpeg-turbo %s (libjpeg API %s)" % (PIL.__version__, features.version("libjpeg_turbo"), features.version("jpg")))
    path = os.path.join(ROOT, "tests", "golden", "jpeg_cases.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(names), "cases,", str(out["decoder"]))


if __name__ == "__main__":
    main()
