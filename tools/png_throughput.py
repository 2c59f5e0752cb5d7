#!/usr/bin/env python
"""Files in, lines out for a directory of PNG files: ctpn/demo_batch.py --decode gpu, whose PNG files go through the
library's host decoder (ctpn_decode_png_files: inflate + row filters on C++ threads, one host-to-device copy per batch), against the
HBM-resident rate of the same box in the same process. No torch in this process (CTPN_NO_TORCH=1): the resident batch is a batch the JPEG
decoder left in device memory.

    CTPN_NO_TORCH=1 python tools/png_throughput.py --images 768 --distinct 96 --out profiles/r04_decode_throughput_png.json
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=768)
    ap.add_argument("--distinct", type=int, default=96)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--threads", type=int, default=0, help="decode threads (0: the rank's host thread budget)")
    ap.add_argument("--out", default=None)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp16", "split", "fp32"], help="arithmetic of the conv stack (the shipped text.yml says split: the parity-grade mode; bf16 is the throughput mode these measurements are quoted in)")
    args = ap.parse_args()
    from PIL import Image
    import io
    import ctpn_amd  # noqa: F401
    from ctpn_amd import _binding as B
    from ctpn_amd.ctpn import demo as D, demo_batch as DB
    from ctpn_amd.lib.networks.factory import get_network
    from ctpn_amd.lib.fast_rcnn.config import cfg_from_file
    from decode_throughput import make_image
    tmp = tempfile.mkdtemp(prefix="ctpn_png_")
    threads = args.threads or B.host_thread_budget(os.cpu_count() or 1, 1, 0)
    out = {"images": args.images, "distinct_images": args.distinct, "height": 600, "width": 900, "host_cpus": os.cpu_count(), "decode_threads": threads}
    try:
        d = os.path.join(tmp, "png")
        os.makedirs(d)
        size = 0
        for i in range(args.images):
            p = os.path.join(d, "img_%04d.png" % i)
            if i < args.distinct:
                Image.fromarray(make_image(i)[:, :, ::-1].copy()).save(p, compress_level=3)
            else:
                shutil.copyfile(os.path.join(d, "img_%04d.png" % (i % args.distinct)), p)
            size += os.path.getsize(p)
        out["mean_file_kb"] = round(size / args.images / 1024, 1)
        out["deflate_backend"] = B.png_backend()
        names = DB.list_images(d)
        # the host decoder alone, all threads
        t0 = time.time()
        for lo in range(0, len(names), args.batch):
            B.decode_png_files(names[lo: lo + args.batch], 600, 900, threads)
        out["decode_only_images_per_s"] = round(len(names) / (time.time() - t0), 1)
        t0 = time.time()
        B.decode_png_files(names[: args.batch], 600, 900, 1)
        out["decode_only_one_thread_images_per_s"] = round(args.batch / (time.time() - t0), 1)
        cfg_from_file(os.path.join(ROOT, "text-detection-ctpn_amd", "ctpn", "text.yml"))
        from ctpn_amd.lib.fast_rcnn.config import cfg
        cfg.TEST.PRECISION = args.precision
        net = get_network("VGGnet_test")
        D.load_weights(net, 0)
        od = os.path.join(tmp, "out")
        DB.run(net, names[: args.batch * 2], od, batch=args.batch, write_images=False, log=lambda *a: None, decode="gpu", decode_threads=threads)
        rates, logs = [], []
        for _ in range(3):
            t0 = time.time()
            DB.run(net, names, od, batch=args.batch, write_images=False, log=logs.append, decode="gpu", decode_threads=threads)
            rates.append(round(len(names) / (time.time() - t0), 1))
        out["demo_batch_png_images_per_s"] = max(rates)
        out["demo_batch_png_runs"] = rates
        out["demo_batch_png_log"] = logs[rates.index(max(rates))]
        # the resident rate: a batch the JPEG decoder leaves in device memory, bench.py's loop
        ctx = net.ctx
        datas = []
        for i in range(args.batch):
            buf = io.BytesIO()
            Image.fromarray(make_image(i)[:, :, ::-1].copy()).save(buf, "JPEG", quality=90)
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
        out["png_file_rate_vs_resident"] = round(out["demo_batch_png_images_per_s"] / out["resident_images_per_s"], 3)
        out["before"] = ("1199 PNG/s end to end with 32 Pillow worker processes (profiles/r04_decode_throughput_procs.json); 2342 with this decoder on "
                         "zlib's inflate and a fresh batch buffer per batch (profiles/r04_decode_throughput_png_first.json)")
        net.close()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    txt = json.dumps(out, indent=1)
    print(txt)
    if args.out:
        open(args.out, "w").write(txt + "\n")


if __name__ == "__main__":
    main()
