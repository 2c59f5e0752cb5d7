"""Generates tests/golden/demo_files.npz from the reference's OWN demo inputs (data/demo/006.jpg .. 009.jpg, 010.png: what `python
ctpn/demo.py` reads, ctpn/demo.py:59,99-100) -- run in the build container, where /root/reference exists:

    python -m oracle.make_demo_golden [/root/reference]

For each file: its bytes (the GPU box has no /root/reference, and these five files ARE the path's reference-held inputs: 1.05 MB), the
shape of what cv2.imread returns for it, the SHA-256 of those pixels and five 32 x 32 windows of them (corners and centre, for a readable
failure). "What cv2.imread returns" = Pillow's decode (libjpeg-turbo / libpng, the decoder families behind OpenCV's imread; cv2 itself is not
in this image) turned by the EXIF orientation (ImageOps.exif_transpose: OpenCV >= 3.1 applies the tag), in BGR order. The files cover what
round 4's device decoder refused: 006 and 009 are 4:4:0 (luma sampled 1 x 2), 008 carries EXIF orientation 6, 007 is 4:2:0, 010 a PNG.
"""
import hashlib
import io
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ["006.jpg", "007.jpg", "008.jpg", "009.jpg", "010.png"]


def cv2_like_bgr(data):
    from PIL import Image, ImageOps
    im = Image.open(io.BytesIO(data))
    if im.format == "JPEG":
        im = ImageOps.exif_transpose(im)
    return np.ascontiguousarray(np.asarray(im.convert("RGB"))[..., ::-1])


def windows(img):
    h, w = img.shape[:2]
    s = 32
    pts = [(0, 0), (0, w - s), (h - s, 0), (h - s, w - s), ((h - s) // 2, (w - s) // 2)]
    return np.stack([img[y:y + s, x:x + s] for y, x in pts])


def main():
    import PIL
    from PIL import features
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    out = {"names": np.array(NAMES)}
    for nm in NAMES:
        data = open(os.path.join(ref, "data", "demo", nm), "rb").read()
        img = cv2_like_bgr(data)
        key = nm.replace(".", "_")
        out["file_" + key] = np.frombuffer(data, np.uint8)
        out["shape_" + key] = np.array(img.shape, np.int64)
        out["sha256_" + key] = np.array(hashlib.sha256(img.tobytes()).hexdigest())
        out["windows_" + key] = windows(img)
        # the reference's own annotated OUTPUT for this file (data/results/<name>, written by draw_boxes, ctpn/demo.py:51-52): its size is
        # cv2.resize's dsize rounding applied twice -- resize_im's factor f, then 1 / f -- to what cv2.imread returned, i.e. evidence the
        # reference holds about (a) the EXIF turn of 008.jpg and (b) the rounding of the output size for five shapes and four factors.
        # res_<stem>.txt is kept too (format evidence only: the trained weights that produced its numbers are not in the tree)
        from PIL import Image
        with Image.open(os.path.join(ref, "data", "results", nm)) as r:
            out["result_hw_" + key] = np.array([r.size[1], r.size[0]], np.int64)
        out["result_txt_" + key] = np.frombuffer(open(os.path.join(ref, "data", "results", "res_%s.txt" % nm.split(".")[0]), "rb").read(), np.uint8)
        print(nm, img.shape, str(out["sha256_" + key])[:16], "reference result image", out["result_hw_" + key].tolist())
    out["decoder"] = np.array("Pillow %s, libjpeg-turbo %s, zlib %s" % (PIL.__version__, features.version("libjpeg_turbo"), features.version("zlib")))
    path = os.path.join(ROOT, "tests", "golden", "demo_files.npz")
    np.savez(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
