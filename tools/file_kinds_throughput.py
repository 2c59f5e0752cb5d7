#!/usr/bin/env python
"""Files in, lines out per FILE KIND: ctpn/demo_batch.py --decode gpu on directories of 600 x 900 files of one kind each
-- sequential 4:2:0 JPEG, progressive 4:2:0 JPEG (same device half, a slower host half), 4:2:2 JPEG, PNG -- against the HBM-resident rate of
the same box in the same process. No torch in this process (CTPN_NO_TORCH=1): the resident batch is one the JPEG decoder left in device memory.

    CTPN_NO_TORCH=1 python tools/file_kinds_throughput.py --images 768 --distinct 64 --out profiles/r05_decode_throughput_kinds.json
"""
import argparse
import io
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

KINDS = {
    "jpg-420": ("jpg", dict(quality=90, subsampling=2)),
    "jpg-420-progressive": ("jpg", dict(quality=90, subsampling=2, progressive=True)),
    "jpg-422": ("jpg", dict(quality=90, subsampling=1)),
    "png": ("png", dict(compress_level=3)),
    # round 5: the layouts of the reference's own data/demo files that used to go to Pillow. exif6: stored 900 x 600 with EXIF orientation 6,
    # shown 600 x 900 (the colour kernel's index map); 440: luma sampled 1 x 2, written by tests/util_jpeg.py's small encoder (Pillow cannot;
    # its plain Huffman tables make the files ~3 x as large as libjpeg's: the host half's worst case), 8 distinct pictures
    "jpg-420-exif6": ("jpg", dict(quality=90, subsampling=2, exif6=True)),
    "jpg-440": ("jpg", dict(custom440=True)),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=768)
    ap.add_argument("--distinct", type=int, default=64)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--kinds", default=",".join(KINDS))
    ap.add_argument("--out", default=None)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp16", "split", "fp32"], help="arithmetic of the conv stack (the shipped text.yml says split: the parity-grade mode; bf16 is the throughput mode these measurements are quoted in)")
    args = ap.parse_args()
    from PIL import Image
    import ctpn_amd  # noqa: F401
    from ctpn_amd import _binding as B
    from ctpn_amd.ctpn import demo as D, demo_batch as DB
    from ctpn_amd.lib.networks.factory import get_network
    from ctpn_amd.lib.fast_rcnn.config import cfg_from_file
    from decode_throughput import make_image
    tmp = tempfile.mkdtemp(prefix="ctpn_kinds_")
    threads = B.host_thread_budget(os.cpu_count() or 1, 1, 0)
    out = {"images": args.images, "distinct_images": args.distinct, "height": 600, "width": 900, "host_cpus": os.cpu_count(), "host_threads": threads,
           "png_deflate_backend": B.png_backend(), "kinds": {}}
    try:
        pics = [Image.fromarray(make_image(i)[:, :, ::-1].copy()) for i in range(args.distinct)]
        dirs = {}
        for kind in args.kinds.split(","):
            ext, kw = KINDS[kind]
            d = os.path.join(tmp, kind)
            os.makedirs(d)
            size = 0
            for i in range(args.images):
                p = os.path.join(d, "img_%04d.%s" % (i, ext))
                if kw.get("custom440"):
                    if i < 8:
                        from util_jpeg import encode_custom
                        import numpy as np
                        open(p, "wb").write(encode_custom(np.asarray(pics[i]), 1, 2, q=6))
                    else:
                        shutil.copyfile(os.path.join(d, "img_%04d.%s" % (i % 8, ext)), p)
                elif kw.get("exif6"):
                    if i < args.distinct:
                        ex = Image.Exif()
                        ex[0x0112] = 6
                        pics[i].transpose(Image.Transpose.ROTATE_90).save(p, quality=kw["quality"], subsampling=kw["subsampling"], exif=ex)
                    else:
                        shutil.copyfile(os.path.join(d, "img_%04d.%s" % (i % args.distinct, ext)), p)
                elif i < args.distinct:
                    pics[i].save(p, **kw)
                else:
                    shutil.copyfile(os.path.join(d, "img_%04d.%s" % (i % args.distinct, ext)), p)
                size += os.path.getsize(p)
            dirs[kind] = d
            out["kinds"][kind] = {"mean_file_kb": round(size / args.images / 1024, 1)}
        cfg_from_file(os.path.join(ROOT, "text-detection-ctpn_amd", "ctpn", "text.yml"))
        from ctpn_amd.lib.fast_rcnn.config import cfg
        cfg.TEST.PRECISION = args.precision
        net = get_network("VGGnet_test")
        D.load_weights(net, 0)
        od = os.path.join(tmp, "out")
        for kind, d in dirs.items():
            names = DB.list_images(d)
            DB.run(net, names[: args.batch * 2], od, batch=args.batch, write_images=False, log=lambda *a: None, decode="gpu", decode_threads=threads)
            rates, logs = [], []
            for _ in range(3):
                t0 = time.time()
                DB.run(net, names, od, batch=args.batch, write_images=False, log=logs.append, decode="gpu", decode_threads=threads)
                rates.append(round(len(names) / (time.time() - t0), 1))
            out["kinds"][kind].update({"images_per_s": max(rates), "runs": rates, "log": logs[rates.index(max(rates))]})
        ctx = net.ctx
        datas = []
        for i in range(args.batch):
            buf = io.BytesIO()
            pics[i % args.distinct].save(buf, "JPEG", quality=90)
            datas.append(buf.getvalue())
        ptr, shape = ctx.decode_jpeg_batch(datas, 600, 900)

        def loop(steps):
            for k in range(steps):
                ctx.detect_submit(device_ptr=ptr, shape=shape, slot=k & 1)
                if k:
                    ctx.detect_collect((k - 1) & 1)
            ctx.detect_collect((steps - 1) & 1)
        loop(3)
        t0 = time.time()
        loop(30)
        out["resident_images_per_s"] = round(args.batch * 30 / (time.time() - t0), 1)
        for v in out["kinds"].values():
            v["vs_resident"] = round(v["images_per_s"] / out["resident_images_per_s"], 3)
        net.close()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    txt = json.dumps(out, indent=1)
    print(txt)
    if args.out:
        open(args.out, "w").write(txt + "\n")


if __name__ == "__main__":
    main()
