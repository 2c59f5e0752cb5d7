#!/usr/bin/env python
"""A loop of ctpn_decode_jpeg_batch calls on one batch of 600x900 JPEG files, for rocprofv3 --kernel-trace --stats (per-kernel time of
jpeg_idct_kernel / jpeg_color_kernel and the H2D copy of the coefficients) and for the host half's wall time.

    python tools/jpeg_loop.py [--batch 32] [--reps 20] [--forward]      # --forward: each decoded batch also goes through detect_submit / collect
"""
import argparse
import io
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--forward", action="store_true")
    args = ap.parse_args()
    from PIL import Image
    import ctpn_amd
    from decode_throughput import make_image
    datas = []
    for i in range(args.batch):
        b = io.BytesIO()
        Image.fromarray(make_image(i)[:, :, ::-1].copy()).save(b, "JPEG", quality=90)
        datas.append(b.getvalue())
    arena = ctpn_amd.make_synthetic_arena(0)
    with ctpn_amd.Context(0, args.batch, 600, 900, "bf16") as ctx:
        ctx.load_weights(arena)
        for _ in range(3):
            ptr, shape = ctx.decode_jpeg_batch(datas, 600, 900)
        ctx.sync()
        host = []
        t0 = time.perf_counter()
        for k in range(args.reps):
            t1 = time.perf_counter()
            ptr, shape = ctx.decode_jpeg_batch(datas, 600, 900)
            host.append(time.perf_counter() - t1)
            if args.forward:
                ctx.detect_submit(device_ptr=ptr, shape=shape, slot=k & 1)
                if k:
                    ctx.detect_collect((k - 1) & 1)
        if args.forward:
            ctx.detect_collect((args.reps - 1) & 1)
        ctx.sync()
        dt = time.perf_counter() - t0
        host.sort()
        print(json.dumps({"batch": args.batch, "reps": args.reps, "forward": args.forward, "images_per_s": round(args.batch * args.reps / dt, 1),
                          "call_ms_median": round(host[len(host) // 2] * 1e3, 3), "call_ms_min": round(host[0] * 1e3, 3),
                          "host_threads": ctx.host_threads(), "mean_file_kb": round(sum(map(len, datas)) / len(datas) / 1024, 1)}))


if __name__ == "__main__":
    main()
