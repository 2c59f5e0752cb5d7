#!/usr/bin/env python
"""End-to-end rate of the batch demo path on FILES: how much of the HBM-resident images/s survives when
the images start as JPEG / PNG files on disk, the way ctpn/demo.py:59 (cv2.imread) gets them.

    python tools/decode_throughput.py --images 512 --out profiles/r04_decode_throughput.json

Writes N synthetic 600x900 "document" images (smooth background, dark text-like strokes: compressible like a scan, not white noise) as JPEG
(quality 90) and as PNG into a scratch directory, then measures, for each format:
  decode_only     images/s of the host decode pool alone (lib/utils/image.py:imread = Pillow, GIL released), per thread count
  demo_batch_gpu  the same with --decode gpu (JPEG): ctpn_decode_jpeg_batch, entropy decoding on the library's host pool + HIP kernels
  demo_batch      images/s of ctpn/demo_batch.py::run end to end (header scan, decode on `threads` host threads one batch ahead, H2D,
                  detect_submit / detect_collect, res_*.txt written by the C++ writer), no annotated images
against `resident`: bench.py's protocol on the same GPU with the uint8 batch already in HBM.
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


def make_image(seed, h=600, w=900):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    base = 200 + 30 * np.sin(xx / (40 + seed % 17)) * np.cos(yy / (55 + seed % 13))
    img = np.repeat(base[:, :, None], 3, axis=2) + rng.normal(0, 3, (h, w, 3))
    for _ in range(40):                                   # text-like dark strokes in rows
        y0 = int(rng.integers(10, h - 30)); x0 = int(rng.integers(10, w - 200))
        hh = int(rng.integers(10, 24)); n = int(rng.integers(4, 18))
        for k in range(n):
            xs = x0 + k * 11
            img[y0:y0 + hh, xs:xs + int(rng.integers(3, 9))] = rng.integers(10, 70)
    return np.clip(img, 0, 255).astype(np.uint8)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=512)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--out", default=None)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp16", "split", "fp32"], help="arithmetic of the conv stack (the shipped text.yml says split: the parity-grade mode; bf16 is the throughput mode these measurements are quoted in)")
    ap.add_argument("--only-procs", action="store_true", help="skip the thread-pool measurements")
    ap.add_argument("--only-gpu", action="store_true", help="JPEG only: demo_batch with worker processes against demo_batch --decode gpu (+ the resident rate)")
    ap.add_argument("--distinct", type=int, default=0, help="encode only this many distinct images and write them under --images names (0 = all distinct)")
    args = ap.parse_args()
    from PIL import Image
    import ctpn_amd
    from ctpn_amd import _binding as B
    from ctpn_amd.ctpn import demo as D, demo_batch as DB
    from ctpn_amd.lib.networks.factory import get_network
    from ctpn_amd.lib.fast_rcnn.config import cfg_from_file
    from ctpn_amd.lib.utils import image as imutil
    from concurrent.futures import ThreadPoolExecutor

    tmp = tempfile.mkdtemp(prefix="ctpn_decode_")
    out = {"images": args.images, "height": 600, "width": 900, "host_cpus": os.cpu_count(), "host_thread_budget": B.host_thread_budget(os.cpu_count() or 1, 1, 0)}
    try:
        dirs = {}
        fmts = ("jpg",) if args.only_gpu else ("jpg", "png")
        for fmt in fmts:
            d = os.path.join(tmp, fmt); os.makedirs(d)
            dirs[fmt] = d
        t0 = time.time()
        sizes = {f: 0 for f in fmts}
        distinct = args.distinct if args.distinct > 0 else args.images
        for i in range(args.images):
            if i < distinct:
                im = Image.fromarray(make_image(i)[:, :, ::-1].copy())
                p = os.path.join(dirs["jpg"], "img_%04d.jpg" % i); im.save(p, quality=90)
                if "png" in dirs:
                    p = os.path.join(dirs["png"], "img_%04d.png" % i); im.save(p, compress_level=3)
            else:
                for fmt in fmts:
                    shutil.copyfile(os.path.join(dirs[fmt], "img_%04d.%s" % (i % distinct, fmt)), os.path.join(dirs[fmt], "img_%04d.%s" % (i, fmt)))
            for fmt in fmts:
                sizes[fmt] += os.path.getsize(os.path.join(dirs[fmt], "img_%04d.%s" % (i, fmt)))
        out["files_written_s"] = round(time.time() - t0, 1)
        out["distinct_images"] = distinct
        out["mean_file_kb"] = {k: round(v / args.images / 1024, 1) for k, v in sizes.items()}
        cfg_from_file(os.path.join(ROOT, "text-detection-ctpn_amd", "ctpn", "text.yml"))
        from ctpn_amd.lib.fast_rcnn.config import cfg
        cfg.TEST.PRECISION = args.precision
        net = get_network("VGGnet_test")
        D.load_weights(net, 0)
        budget = out["host_thread_budget"]
        res = {}
        if args.only_gpu:
            args.only_procs = True
        for fmt in fmts:
            names = DB.list_images(dirs[fmt])
            r = {"decode_only_images_per_s": {}, "demo_batch_images_per_s": {}}
            for th in ([] if args.only_procs else sorted({1, 8, budget})):
                with ThreadPoolExecutor(max_workers=th) as pool:
                    t0 = time.time()
                    list(pool.map(imutil.imread, names))
                    r["decode_only_images_per_s"][str(th)] = round(len(names) / (time.time() - t0), 1)
            for th in ([] if args.only_procs else sorted({8, budget})):
                od = os.path.join(tmp, "out_%s_%d" % (fmt, th))
                DB.run(net, names[: args.batch * 2], od, batch=args.batch, write_images=False, log=lambda *a: None, decode_threads=th)     # warm-up
                t0 = time.time()
                DB.run(net, names, od, batch=args.batch, write_images=False, log=lambda *a: None, decode_threads=th)
                r["demo_batch_images_per_s"][str(th)] = round(len(names) / (time.time() - t0), 1)
            r["demo_batch_procs_images_per_s"] = {}
            for pr in ([budget] if args.only_procs else sorted({8, budget})):
                od = os.path.join(tmp, "outp_%s_%d" % (fmt, pr))
                pool = DB.decode_pool(pr)                                     # warm worker processes (a service keeps them; start-up is ~1.5 s)
                try:
                    DB.run(net, names[: args.batch * 2], od, batch=args.batch, write_images=False, log=lambda *a: None, decode_pool=pool)
                    t0 = time.time()
                    DB.run(net, names, od, batch=args.batch, write_images=False, log=lambda *a: None, decode_pool=pool)
                    r["demo_batch_procs_images_per_s"][str(pr)] = round(len(names) / (time.time() - t0), 1)
                finally:
                    pool.shutdown()
            if fmt == "jpg":
                # decode + resize_im on the device (ctpn_decode_jpeg_batch): entropy decoding on the library's C++ pool, the rest as HIP kernels
                od = os.path.join(tmp, "outg_%s" % fmt)
                DB.run(net, names[: args.batch * 2], od, batch=args.batch, write_images=False, log=lambda *a: None, decode="gpu")
                rates, logs = [], []
                for _ in range(5):
                    t0 = time.time()
                    DB.run(net, names, od, batch=args.batch, write_images=False, log=logs.append, decode="gpu")
                    rates.append(round(len(names) / (time.time() - t0), 1))
                r["demo_batch_gpu_decode_images_per_s"] = max(rates)
                r["demo_batch_gpu_decode_runs"] = rates
                r["demo_batch_gpu_decode_log"] = logs[rates.index(max(rates))]
                # the decoder alone: files already in memory, no detector
                datas = [open(nm, "rb").read() for nm in names[: args.batch * 4]]
                ctx = net.ctx
                for k in range(2):
                    ctx.decode_jpeg_batch(datas[: args.batch], 600, 900)
                ctx.sync()
                t0 = time.time()
                reps = 12
                for k in range(reps):
                    lo = (k % 4) * args.batch
                    ptr, shape = ctx.decode_jpeg_batch(datas[lo: lo + args.batch], 600, 900)
                ctx.jpeg_batch_fetch(ptr, shape)
                r["decode_only_gpu_images_per_s"] = round(reps * args.batch / (time.time() - t0), 1)
            res[fmt] = r
        out["formats"] = res
        # the HBM-resident rate on this box, same batch, bench.py's protocol
        import torch
        imgs = torch.from_numpy(np.stack([make_image(i) for i in range(args.batch)])).cuda()
        ctx = net.ctx
        for k in range(3):
            ctx.detect_submit(device_ptr=imgs.data_ptr(), shape=(args.batch, 600, 900), slot=k & 1)
            if k:
                ctx.detect_collect((k - 1) & 1)
        ctx.detect_collect(0)
        torch.cuda.synchronize()
        steps = 30
        t0 = time.time()
        for k in range(steps):
            ctx.detect_submit(device_ptr=imgs.data_ptr(), shape=(args.batch, 600, 900), slot=k & 1)
            if k:
                ctx.detect_collect((k - 1) & 1)
        ctx.detect_collect((steps - 1) & 1)
        torch.cuda.synchronize()
        out["resident_images_per_s"] = round(args.batch * steps / (time.time() - t0), 1)
        best = max(max(list(v["demo_batch_images_per_s"].values()) + list(v["demo_batch_procs_images_per_s"].values()) + [v.get("demo_batch_gpu_decode_images_per_s", 0)]) for v in res.values())
        out["best_file_rate_vs_resident"] = round(best / out["resident_images_per_s"], 3)
        net.close()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    txt = json.dumps(out, indent=1)
    print(txt)
    if args.out:
        open(args.out, "w").write(txt + "\n")


if __name__ == "__main__":
    main()
