#!/usr/bin/env python
"""images/s of the CTPN inference hot path at 600x900 on N MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = one pass of the whole path (uint8 images resident in HBM -> conv stack -> BiLSTM -> heads -> proposal
layer -> connector front end on device -> text lines on the host) over one batch of 32 synthetic 600x900 images
per GPU: BASELINE.json configs[2] (bf16 MFMA conv stack + fp32 BiLSTM) -- the configuration the metric's scaling
curve is quoted on (configs[3] = 32 images per GPU x N). Weak scaling: per-GPU work is fixed, ranks never exchange
data on the path; the weight arena is broadcast once (RCCL) before the timed region.

Rank 0 prints ONE JSON line. Extra objects: "roofline" for the dominant kernel (implicit-GEMM conv, MFMA-bound),
measured live with hipEvents on the ctx stream over the timed region, and "cpu_baseline": the oracle (CPU port
of the reference path) timed on this host's cores on a bounded sample of the same workload (N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CONV_GFLOP_PER_IMAGE_600x900 = 337.264  # 13 implicit-GEMM convs: 339.130 minus conv1_1's 1.866 (direct kernel)
PEAK = {"bf16": 2500.0, "fp32": 157.3}   # dense MFMA TFLOP/s, MI355X_MICROARCH.md


def cpu_baseline(arena, h, w, n_images, mode):
    """The oracle end to end on the host: torch-CPU fp32 forward + numpy proposal layer / NMS / connector.
    Protocol of BASELINE.md section 2.3: 2 warm-up images (mirrors ctpn/demo.py:95-97), then the MEDIAN of >= 5 timed images,
    with the per-stage split (conv stack / BiLSTM + heads / proposal layer + NMS / connector)."""
    import torch
    import ctpn_amd
    from oracle import network as N
    from oracle import postproc as P
    torch.set_grad_enabled(False)
    wts = ctpn_amd.arena_views(arena)
    info = np.array([h, w, 1.0], np.float32)

    def one(seed):
        t = [time.perf_counter()]
        img = ctpn_amd.weights.synthetic_images(1, h, w, seed)
        x = N.image_blob(img)
        for name in N.CONVS:
            x = N.conv3x3_relu(x, wts[name + "/weights"], wts[name + "/biases"])
            if name in N.POOL_AFTER:
                x = N.maxpool2x2(x)
        t.append(time.perf_counter())
        fc = N.dense(N.bilstm(x, wts), wts["lstm_o/weights"], wts["lstm_o/biases"])
        bbox = N.dense(fc, wts["rpn_bbox_pred/weights"], wts["rpn_bbox_pred/biases"])
        cls = N.pair_softmax(N.dense(fc, wts["rpn_cls_score/weights"], wts["rpn_cls_score/biases"]))
        t.append(time.perf_counter())
        rois = P.proposal_layer(cls, bbox, info)
        t.append(time.perf_counter())
        P.text_detect(rois[:, 1:5], rois[:, 0], (h, w), mode)
        t.append(time.perf_counter())
        return [t[i + 1] - t[i] for i in range(4)] + [t[4] - t[0]]

    for s in (1, 2):      # 2 warm-ups
        one(s)
    n_images = max(5, n_images)
    rows = np.array([one(1 + i) for i in range(n_images)])
    med = np.median(rows, axis=0)
    return {"value": round(1.0 / med[4], 4), "unit": "images/s", "cores": int(torch.get_num_threads()),
            "host_cpus": os.cpu_count(), "kind": "port",
            "seconds_per_image_median": round(float(med[4]), 4),
            "stages_s_median": {"conv_stack": round(float(med[0]), 4), "bilstm_heads": round(float(med[1]), 4),
                                "proposal_nms": round(float(med[2]), 4), "connector": round(float(med[3]), 4)},
            "sample": "median of %d synthetic %dx%d images after 2 warm-up images, one image at a time (the reference is batch-1); "
                      "oracle/network.py (torch CPU fp32, %d threads) + oracle/postproc.py (numpy, 1 thread)" % (n_images, h, w, torch.get_num_threads())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU per step")
    ap.add_argument("--height", type=int, default=600)
    ap.add_argument("--width", type=int, default=900)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--mode", default="H", choices=["H", "O"])
    ap.add_argument("--cpu-images", type=int, default=6, help="images in the CPU-baseline sample (0 disables)")
    ap.add_argument("--host-images", action="store_true",
                    help="feed host uint8 batches instead of HBM-resident ones (PCIe-inclusive rate; never the headline value)")
    ap.add_argument("--stage-events", default="after", choices=["after", "inline", "off"],
                    help="per-stage hipEvent pairs: in an extra untimed pass after the timed region (default), inside it, or not at all")
    ap.add_argument("--lstm-split", action="store_true",
                    help="BiLSTM recurrence on split-bf16 MFMAs (fp32-class accuracy) instead of the exact-fp32 MFMA kernel; not the BASELINE config")
    ap.add_argument("--pinned", action="store_true", help="with --host-images: page-locked host buffer (truly asynchronous H2D)")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "r02_pmc.json" if os.path.exists(os.path.join(ROOT, "profiles", "r02_pmc.json")) else "r01_pmc.json"),
                    help="PMC summary (tools/pmc_summary.py over separate rocprofv3 --pmc passes) that fills roofline.traffic")
    args = ap.parse_args()

    import torch
    import ctpn_amd
    from ctpn_amd import dist as D

    if args.lstm_split:
        os.environ["CTPN_LSTM_SPLIT"] = "1"
    rank, local_rank, world = D.env_world()
    if world > 1:
        D.init_process_group("nccl")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    B, H, W = args.batch, args.height, args.width

    # weights: rank 0 builds the arena, ONE broadcast over RCCL/xGMI, each rank packs from its HBM copy
    arena = ctpn_amd.make_synthetic_arena(0) if rank == 0 else None
    t_b0 = time.time()
    arena_dev = D.broadcast_arena(arena, dev, src=0)
    torch.cuda.synchronize()
    t_bcast = time.time() - t_b0
    ctx = ctpn_amd.Context(local_rank, B, H, W, args.precision)
    ctx.load_weights_device(arena_dev.data_ptr())

    # this rank's shard of the global image list (seeds 1 .. world*B), resident in HBM before the timed region
    lo, hi = D.shard_range(world * B, rank, world)
    imgs = torch.from_numpy(np.stack([np.random.default_rng(1 + i).integers(0, 256, size=(H, W, 3), dtype=np.uint8)
                                      for i in range(lo, hi)])).to(dev)
    torch.cuda.synchronize()
    shape = (hi - lo, H, W)
    imgs_host = None
    if args.host_images:
        ht = imgs.cpu()
        imgs_host = (ht.pin_memory() if args.pinned else ht).numpy()

    def run(k_steps):
        """k_steps passes of the hot path, software-pipelined over the ctx's two slots: the device part of step k+1
        (ctpn_detect_submit) is enqueued before the host part of step k (ctpn_detect_collect) runs. Every step is
        fully collected before this returns."""
        out = None
        for k in range(k_steps):
            if args.host_images:
                ctx.detect_submit(images=imgs_host, slot=k & 1)
            else:
                ctx.detect_submit(device_ptr=imgs.data_ptr(), shape=shape, slot=k & 1)
            if k > 0:
                out = ctx.detect_collect((k - 1) & 1, mode=args.mode, line_capacity=512)
        if k_steps > 0:
            out = ctx.detect_collect((k_steps - 1) & 1, mode=args.mode, line_capacity=512)
        return out

    lines = run(args.warmup)
    # Timed region: ONE hipEvent pair per step around the 13 conv3x3 launches (the roofline kernel) on the ctx stream; a pair
    # around every stage (42 records per step) costs ~0.2 ms of bubbles per step, so the per-stage split is taken in a
    # separate, untimed pass afterwards.
    ctx.profile_enable(2 if args.stage_events == "after" else (True if args.stage_events == "inline" else False))
    ctx.profile_reset()
    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    lines = run(args.steps)
    torch.cuda.synchronize()
    D.barrier()
    elapsed_local = time.perf_counter() - t0
    elapsed = D.max_over_ranks(elapsed_local, dev)
    per_rank = D.gather_over_ranks([elapsed_local / args.steps * 1e3, t_bcast * 1e3, ctx.host_threads()], dev)
    prof = ctx.profile_read()
    ctx.profile_enable(False)
    stage_steps = args.steps
    if args.stage_events == "after":
        conv_timed = prof["conv_gemm"]
        ctx.profile_enable(True)
        ctx.profile_reset()
        stage_steps = min(args.steps, 5)
        run(stage_steps)
        torch.cuda.synchronize()
        prof = ctx.profile_read()
        ctx.profile_enable(False)
        prof_stage_conv = prof["conv_gemm"]
        prof["conv_gemm"] = conv_timed          # the roofline uses the timed region's measurement
    else:
        prof_stage_conv = prof["conv_gemm"]

    if rank == 0:
        total_images = world * B * args.steps
        cg = prof["conv_gemm"]
        achieved = cg["work"] / (cg["ms"] * 1e-3) / 1e12 if cg["ms"] > 0 else 0.0
        traffic = None
        if args.traffic_json and os.path.exists(args.traffic_json):
            # measured in separate rocprofv3 --pmc passes of this same command (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE,
            # KiB -> bytes), average over the 13 conv launches of a step; only valid for the default workload
            if (B, H, W, args.precision) == (32, 600, 900, "bf16"):
                traffic = json.load(open(args.traffic_json)).get("hbm_bytes_per_launch")
        out = {
            "metric": "images/sec at 600x900 (VGG16-CTPN inference)",
            "value": round(total_images / elapsed, 2),
            "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic" + ((" (host-resident%s, H2D copy inside the timed region)" % (", page-locked" if args.pinned else "")) if args.host_images else ""),
            "config": {"workload": ("batch=%d at %dx%d per GPU, %s MFMA conv stack + fp32 BiLSTM%s + HIP proposal/NMS + text lines (%s); "
                                    "BASELINE.json configs[2], sharded as configs[3] for N>1") % (
                                        B, H, W, args.precision, " (recurrent product on split-bf16 MFMAs)" if os.environ.get("CTPN_LSTM_SPLIT") == "1" else "", args.mode),
                       "images_per_gpu": B, "global_batch": world * B, "height": H, "width": W,
                       "parallelism": "data-parallel replicas, %d rank(s), one weight broadcast (%.1f ms), no per-batch collective" % (world, t_bcast * 1e3),
                       "weights": "seeded random init (ctpn_amd.make_synthetic_arena(0)); no trained checkpoint exists in the reference tree",
                       "lines_rank0_last_step": int(sum(len(l) for l in lines)),
                       "host_threads_per_rank": int(per_rank[0][2])},
            "per_rank": {"ms_per_step": [round(r[0], 3) for r in per_rank], "weight_broadcast_ms": [round(r[1], 1) for r in per_rank]},
            "roofline": {"kernel": "ctpn::conv3x3_wr_kernel x2 (conv1_2, conv2_1: weights in registers) + ctpn::conv3x3_p_kernel x11 (tap-reuse MFMA conv3x3 + bias + ReLU "
                                   "(+ 2x2 max-pool)), 13 launches per step (+ 5 conv3x3_edge_kernel launches for ragged tile columns, concurrent with their layers); one hipEvent pair per step around them, gaps included", "bound": "mfma",
                         "achieved": round(achieved, 2), "peak": PEAK[args.precision], "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK[args.precision], 4), "traffic": traffic,
                         "launches": cg["launches"], "avg_launch_ms": round(cg["ms"] / max(cg["launches"], 1), 4),
                         "flops_per_launch_avg": cg["work"] / max(cg["launches"], 1),
                         "flops_per_image": CONV_GFLOP_PER_IMAGE_600x900 * 1e9 if (H, W) == (600, 900) else None},
            "stages_ms_per_step": {k: round((prof_stage_conv["ms"] if k == "conv_gemm" else v["ms"]) / stage_steps, 4) for k, v in prof.items()},
            "stage_events": args.stage_events,
        }
        if world == 1 and args.cpu_images > 0:
            out["cpu_baseline"] = cpu_baseline(arena, H, W, args.cpu_images, args.mode)
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
