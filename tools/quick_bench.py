#!/usr/bin/env python
"""Same-box A/B of libraries / precisions / per-ctx options WITHOUT torch in the process: a fresh GPU box pays one to two minutes of its
budget for the first `import torch`, bench.py's timing loop does not need it. Every variant is a ctx of its own; the variants are timed
round-robin (variant 0, 1, ..., 0, 1, ...) so that clock and box drift hit all of them alike. The batch is resident in HBM: it is what the
JPEG decoder leaves behind for a batch of synthetic noise images (quality 100, 4:4:4 -- statistically the benchmark's images, not byte-equal to
them; for the JUDGED number use bench.py).

    CTPN_NO_TORCH=1 python tools/quick_bench.py --variant "" --variant "conv1_fuse=0" --variant "precision=fp16" [--steps 40] [--rounds 3]
    CTPN_NO_TORCH=1 python tools/quick_bench.py --variant "lib=text-detection-ctpn_amd/libctpn_hip_exp.so"      (needs its own process: one library per process)

A variant is a space-separated list of NAME=VALUE: `precision=...`, `batch=...`, or any per-ctx option of ctpn_set_option. Prints one line per
variant and round, then the per-variant medians as JSON (images/s, ms per step, conv-stack TFLOP/s from the ctx's own hipEvents, stage split).
"""
import argparse
import io
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", action="append", default=[], help='e.g. "" (defaults), "conv1_fuse=0", "precision=fp16 lstm_split=0"')
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--height", type=int, default=600)
    ap.add_argument("--width", type=int, default=900)
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--mode", default="H")
    ap.add_argument("--stages", action="store_true", help="one more untimed pass per variant with an event pair around every stage")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    variants = args.variant or [""]
    for v in variants:
        for kv in v.split():
            if kv.startswith("lib="):
                os.environ["CTPN_LIB_PATH"] = os.path.join(ROOT, kv[4:]) if not os.path.isabs(kv[4:]) else kv[4:]
    from PIL import Image
    import ctpn_amd
    B, H, W = args.batch, args.height, args.width
    arena = ctpn_amd.make_synthetic_arena(0)
    files = []
    for i in range(B):
        buf = io.BytesIO()
        Image.fromarray(np.random.default_rng(1 + i).integers(0, 256, size=(H, W, 3), dtype=np.uint8)).save(buf, "JPEG", quality=100, subsampling=0)
        files.append(buf.getvalue())
    ctxs = []
    for v in variants:
        kv = dict(x.split("=", 1) for x in v.split())
        kv.pop("lib", None)
        prec = kv.pop("precision", args.precision)
        b = int(kv.pop("batch", B))
        ctx = ctpn_amd.Context(0, b, H, W, prec, options={k: int(x) for k, x in kv.items()})
        ctx.load_weights(arena)
        ptr, shape = ctx.decode_jpeg_batch(files[:b], H, W)          # stays valid: this ctx decodes nothing else
        ctxs.append((v or "(defaults)", prec, b, ctx, ptr, shape))

    def run(ctx, ptr, shape, steps):
        for k in range(steps):
            ctx.detect_submit(device_ptr=ptr, shape=shape, slot=k & 1)
            if k:
                ctx.detect_collect((k - 1) & 1, mode=args.mode)
        return ctx.detect_collect((steps - 1) & 1, mode=args.mode)

    rows = {name: [] for name, *_ in ctxs}
    for name, prec, b, ctx, ptr, shape in ctxs:
        run(ctx, ptr, shape, args.warmup)
    for r in range(args.rounds):
        for name, prec, b, ctx, ptr, shape in ctxs:
            ctx.profile_enable(2)
            ctx.profile_reset()
            ctx.sync()
            t0 = time.perf_counter()
            lines = run(ctx, ptr, shape, args.steps)
            ctx.sync()
            dt = time.perf_counter() - t0
            cg = ctx.profile_read()["conv_gemm"]
            ctx.profile_enable(False)
            tf = cg["work"] / (cg["ms"] * 1e-3) / 1e12 if cg["ms"] > 0 else 0.0
            row = {"images_per_s": b * args.steps / dt, "ms_per_step": dt / args.steps * 1e3, "conv_stack_tflops": tf,
                   "conv_stack_ms_per_step": cg["ms"] / args.steps, "lines_last_step": int(sum(len(x) for x in lines))}
            rows[name].append(row)
            print("round %d  %-40s %8.1f images/s  %7.3f ms/step  conv stack %7.1f TF  (%.3f ms)" % (
                r, name, row["images_per_s"], row["ms_per_step"], tf, row["conv_stack_ms_per_step"]), flush=True)
    out = {"batch": B, "height": H, "width": W, "steps": args.steps, "rounds": args.rounds, "variants": {}}
    for name, prec, b, ctx, ptr, shape in ctxs:
        med = {k: float(np.median([x[k] for x in rows[name]])) for k in rows[name][0]}
        med["precision"], med["batch"] = prec, b
        if args.stages:
            ctx.profile_enable(True)
            ctx.profile_reset()
            n = min(args.steps, 5)
            run(ctx, ptr, shape, n)
            ctx.sync()
            med["stages_ms_per_step"] = {k: round(v["ms"] / n, 4) for k, v in ctx.profile_read().items()}
            ctx.profile_enable(False)
        out["variants"][name] = med
        ctx.close()
    txt = json.dumps(out, indent=1)
    print(txt)
    if args.out:
        open(args.out, "w").write(txt + "\n")


if __name__ == "__main__":
    main()
