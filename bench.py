#!/usr/bin/env python
"""images/s of the CTPN inference hot path at 600x900 on N MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 100 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = one pass of the whole path (uint8 images resident in HBM -> conv stack -> BiLSTM -> heads -> proposal
layer -> connector front end on device -> text lines on the host) over one batch of 32 synthetic 600x900 images
per GPU: BASELINE.json configs[2] (bf16 MFMA conv stack + fp32 BiLSTM) -- the configuration the metric's scaling
curve is quoted on (configs[3] = 32 images per GPU x N). Weak scaling: per-GPU work is fixed, ranks never exchange
data on the path; the weight arena is broadcast once over RCCL (ctpn_broadcast_weights_rank of the C ABI) before the
timed region. torch.distributed (gloo) is rendezvous / barrier / MAX-of-a-scalar plumbing only.

Rank 0 prints ONE JSON line. Extra objects:
  roofline       the dominant kernel (implicit-GEMM conv, MFMA-bound), measured live with hipEvents on the ctx stream over the
                 timed region;
  cpu_baseline   the oracle (CPU port of the reference path) timed on this host's cores on a bounded sample of the same workload
                 (N = 1 only);
  accuracy       the device outputs of the SAME sample images against that oracle run (cls_prob, rois, text lines);
  other_configs  after the timed region (never inside it), N = 1 only: BASELINE.json configs[4] (8 x 1280x1920, DETECT_MODE=O), the fp32
                 correctness-gate path at batch 8 and 32, the split-precision mode (parity-grade, three bf16 MFMAs per product) and the fp16 mode at
                 batch 32 -- each with its accuracy against the oracle --, batch-1 throughput (pipelined) and the single-image SYNCHRONOUS latency (the reference's calling convention), and the PCIe-inclusive rate (page-locked and pageable).
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
CONV1_1_GFLOP_PER_IMAGE_600x900 = 1.866  # inside conv1_2's launch when the ctx computes conv1_1 in its window stage (16-bit modes, uint8 feed)
PEAK = {"bf16": 2500.0, "fp16": 2500.0, "split": 2500.0, "fp32": 157.3}   # dense MFMA TFLOP/s of the opcode each mode issues, MI355X_MICROARCH.md
MFMA_PER_PRODUCT = {"bf16": 1, "fp16": 1, "split": 3, "fp32": 1}        # split precision spends three bf16 MFMAs per algorithmic product


# ---------------------------------------------------------------------------------------------------------------
# matching helpers of the accuracy object (pure numpy; one-to-one greedy pairing within a tolerance)
# ---------------------------------------------------------------------------------------------------------------
def _match_frac(got, ref, cols, px_tol, score_col=None, score_tol=None):
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    if got.shape[0] == 0:
        return 1.0
    if ref.shape[0] == 0:
        return 0.0
    used = np.zeros(ref.shape[0], bool)
    hit = 0
    for g in got:
        ok = (np.abs(ref[:, cols] - g[cols]).max(axis=1) <= px_tol) & ~used
        if score_col is not None:
            ok &= np.abs(ref[:, score_col] - g[score_col]) <= score_tol
        if ok.any():
            used[np.argmax(ok)] = True
            hit += 1
    return hit / float(got.shape[0])


def _hull_iou_frac(got, ref, thr=0.7):
    got = np.asarray(got, np.float64).reshape(-1, 9)
    ref = np.asarray(ref, np.float64).reshape(-1, 9)
    if got.shape[0] == 0:
        return 1.0
    if ref.shape[0] == 0:
        return 0.0
    hull = lambda r: np.stack([r[:, 0:8:2].min(1), r[:, 1:8:2].min(1), r[:, 0:8:2].max(1), r[:, 1:8:2].max(1)], 1)
    a, b = hull(got), hull(ref)
    hit = 0
    for g in a:
        iw = np.maximum(0, np.minimum(g[2], b[:, 2]) - np.maximum(g[0], b[:, 0]) + 1)
        ih = np.maximum(0, np.minimum(g[3], b[:, 3]) - np.maximum(g[1], b[:, 1]) + 1)
        inter = iw * ih
        union = (g[2] - g[0] + 1) * (g[3] - g[1] + 1) + (b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1) - inter
        hit += bool((inter / union > thr).any())
    return hit / float(a.shape[0])


def accuracy_against(oracle_out, dev_cls, dev_rois, dev_lines, geom=None):
    """oracle_out: list of (cls_prob, rois, lines[, bbox_pred]) per sample image from the cpu_baseline leg; dev_*: the device's outputs for
    the same images. north_star's bar (1e-3 on scores, +-1 px on boxes) is the fp32 path's; the bf16 path is reported against the same
    oracle. geom = (h, w, mode) and a 4-tuple oracle_out: also `text_line_match_frac_1px_given_device_scores` -- the device's lines against
    the ORACLE post-processing of [device cls_prob, ORACLE bbox_pred]. Where that is 1.0 and text_line_match_frac_1px is not, every line
    outside 1 px moved through score differences alone (cls_prob_max_abs_diff: an order flip between two overlapping proposals whose
    scores are an fp32 ulp or two apart -- tests/test_gpu_round6.py::test_config5_geometry_on_the_bench_sample_seeds)."""
    d = [np.abs(dev_cls[i] - o[0]) for i, o in enumerate(oracle_out)]
    extra = {}
    if geom is not None and oracle_out and len(oracle_out[0]) > 3:
        from oracle import postproc as P
        h, w, mode = geom
        info = np.array([h, w, 1.0], np.float32)
        fr = []
        unmatched = swaps = flips = 0
        for i, o in enumerate(oracle_out):
            hyb = P.proposal_layer(dev_cls[i][None], o[3][None], info)
            fr.append(_match_frac(dev_lines[i], P.text_detect(hyb[:, 1:5], hyb[:, 0], (h, w), mode), slice(0, 8), 1.0))
            # device rois without an oracle partner: swaps at the post_nms_topN cut (the device roi IS in the oracle's list continued past the cut
            # and its score is within 4 fp32 ulps of the oracle's last kept score) or something else?
            got, ref = np.asarray(dev_rois[i], np.float64), np.asarray(o[1], np.float64)
            used = np.zeros(len(ref), bool)
            miss = []
            for k, g in enumerate(got):
                ok = (np.abs(ref[:, 1:5] - g[1:5]).max(axis=1) <= 1.0) & (np.abs(ref[:, 0] - g[0]) <= 1e-3) & ~used
                if ok.any():
                    used[np.argmax(ok)] = True
                else:
                    miss.append(k)
            unmatched += len(miss)
            if miss and len(ref):
                ext = np.asarray(P.proposal_layer(o[0][None], o[3][None], info, post_nms_topn=len(ref) + 32), np.float64)
                for k in miss:
                    in_ext = ((np.abs(ext[:, 1:5] - got[k, 1:5]).max(axis=1) <= 1.0) & (np.abs(ext[:, 0] - got[k, 0]) <= 1e-3)).any()
                    cut = bool(in_ext and abs(got[k, 0] - ref[-1, 0]) <= 4 * 2.0 ** -24)
                    swaps += int(cut)
                    if not cut:
                        # the other knife edge: an oracle roi of (almost) the same score that overlaps this one above the NMS threshold -- the two
                        # suppress each other, and which of them is kept is decided by which ranks first
                        g = got[k]
                        iw = np.maximum(0.0, np.minimum(g[3], ref[:, 3]) - np.maximum(g[1], ref[:, 1]) + 1)
                        ih = np.maximum(0.0, np.minimum(g[4], ref[:, 4]) - np.maximum(g[2], ref[:, 2]) + 1)
                        inter = iw * ih
                        iou = inter / ((g[3] - g[1] + 1) * (g[4] - g[2] + 1) + (ref[:, 3] - ref[:, 1] + 1) * (ref[:, 4] - ref[:, 2] + 1) - inter)
                        flips += int(((iou > 0.7) & (np.abs(ref[:, 0] - g[0]) <= 4 * 2.0 ** -24)).any())
        extra["text_line_match_frac_1px_given_device_scores"] = float(np.mean(fr))
        extra["rois_without_partner"] = int(unmatched)
        extra["rois_without_partner_that_are_topn_cut_swaps"] = int(swaps)
        extra["rois_without_partner_that_are_nms_tie_flips"] = int(flips)       # (scores within 4 fp32 ulps, IoU > 0.7 with the oracle roi that took its place)
    return {**extra, **{
        "images": len(oracle_out),
        "cls_prob_max_abs_diff": float(max(x.max() for x in d)),
        "cls_prob_mean_abs_diff": float(np.mean([x.mean() for x in d])),
        "roi_match_frac_1px_1e-3": float(np.mean([_match_frac(dev_rois[i], o[1], slice(1, 5), 1.0, 0, 1e-3) for i, o in enumerate(oracle_out)])),
        "roi_match_frac_1px_1e-2": float(np.mean([_match_frac(dev_rois[i], o[1], slice(1, 5), 1.0, 0, 1e-2) for i, o in enumerate(oracle_out)])),
        "text_line_match_frac_1px": float(np.mean([_match_frac(dev_lines[i], o[2], slice(0, 8), 1.0) for i, o in enumerate(oracle_out)])),
        "text_line_match_frac_iou0.7": float(np.mean([_hull_iou_frac(dev_lines[i], o[2]) for i, o in enumerate(oracle_out)])),
        "text_lines_device": int(sum(len(l) for l in dev_lines)), "text_lines_oracle": int(sum(len(o[2]) for o in oracle_out)),
        "oracle": "oracle/network.py (torch CPU fp32) + oracle/postproc.py on the same images (the cpu_baseline sample); a roi / line "
                  "'matches' if a one-to-one oracle partner lies within the tolerance",
    }}


def cpu_baseline(arena, h, w, n_images, mode):
    """The oracle end to end on the host: torch-CPU fp32 forward + numpy proposal layer / NMS / connector.
    Protocol of BASELINE.md section 2.3: 2 warm-up images (mirrors ctpn/demo.py:95-97), then the MEDIAN of >= 5 timed images,
    with the per-stage split (conv stack / BiLSTM + heads / proposal layer + NMS / connector). Returns (baseline object, the
    oracle's outputs per timed image: the accuracy object compares the device against them)."""
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
        lines = P.text_detect(rois[:, 1:5], rois[:, 0], (h, w), mode)
        t.append(time.perf_counter())
        return [t[i + 1] - t[i] for i in range(4)] + [t[4] - t[0]], (cls[0], rois, lines, bbox[0])

    for s in (1, 2):      # 2 warm-ups
        one(s)
    n_images = max(5, n_images)
    runs = [one(1 + i) for i in range(n_images)]
    rows = np.array([r[0] for r in runs])
    med = np.median(rows, axis=0)
    base = {"value": round(1.0 / med[4], 4), "unit": "images/s", "cores": int(torch.get_num_threads()),
            "host_cpus": os.cpu_count(), "kind": "port",
            "seconds_per_image_median": round(float(med[4]), 4),
            "stages_s_median": {"conv_stack": round(float(med[0]), 4), "bilstm_heads": round(float(med[1]), 4),
                                "proposal_nms": round(float(med[2]), 4), "connector": round(float(med[3]), 4)},
            "sample": "median of %d synthetic %dx%d images (seeds 1..%d) after 2 warm-up images, one image at a time (the reference is batch-1); "
                      "oracle/network.py (torch CPU fp32, %d threads) + oracle/postproc.py (numpy, 1 thread)" % (n_images, h, w, n_images, torch.get_num_threads())}
    return base, [r[1] for r in runs]


def oracle_outputs(arena, h, w, n_images, mode):
    """(cls_prob, rois, lines) of the oracle for the images of seeds 1..n at another geometry (config 5): the accuracy object's reference,
    untimed."""
    import torch
    import ctpn_amd
    from oracle import network as N
    from oracle import postproc as P
    torch.set_grad_enabled(False)
    wts = ctpn_amd.arena_views(arena)
    info = np.array([h, w, 1.0], np.float32)
    out = []
    for i in range(n_images):
        ref = N.forward(ctpn_amd.weights.synthetic_images(1, h, w, 1 + i), wts, keep=set())
        rois = P.proposal_layer(ref["rpn_cls_prob_reshape"], ref["rpn_bbox_pred"], info)
        out.append((ref["rpn_cls_prob_reshape"][0], rois, P.text_detect(rois[:, 1:5], rois[:, 0], (h, w), mode), ref["rpn_bbox_pred"][0]))
    return out


def device_sample_outputs(ctpn_amd, arena, precision, h, w, n_images, mode):
    """cls_prob, rois and text lines of the product path for the cpu_baseline sample images (seeds 1..n), one synchronous batch."""
    imgs = ctpn_amd.weights.synthetic_images(n_images, h, w, 1)
    with ctpn_amd.Context(0, n_images, h, w, precision) as ctx:
        ctx.load_weights(arena)
        lines, rois = ctx.detect(imgs, mode=mode, want_rois=True, line_capacity=1024)
        cls = ctx.get_tensor("rpn_cls_prob_reshape")
    return cls, rois, lines


def run_config(ctpn_amd, torch, dev, ctx, imgs, shape, steps, warmup, mode, host_images=None, stage_events="after", sync=None):
    """warmup untimed + exactly `steps` timed passes of the hot path on `ctx`, software-pipelined over the ctx's two slots: the device
    part of step k+1 (ctpn_detect_submit) is enqueued before the host part of step k (ctpn_detect_collect) runs; every step is fully
    collected before the clock stops. sync(): barrier + torch.cuda.synchronize() on both sides of the timed region.
    Returns (elapsed seconds of this rank, profile dict, per-stage profile dict, stage steps, (lines, rois) of the last timed step)."""
    def run(k_steps):
        out = None
        for k in range(k_steps):
            if host_images is not None:
                ctx.detect_submit(images=host_images, slot=k & 1)
            else:
                ctx.detect_submit(device_ptr=imgs.data_ptr(), shape=shape, slot=k & 1)
            if k > 0:
                out = ctx.detect_collect((k - 1) & 1, mode=mode, line_capacity=512)
        if k_steps > 0:
            # the LAST step's rois come back too (they are in the slot's one D2H copy anyway): the caller checks the timed batch's output
            out = ctx.detect_collect((k_steps - 1) & 1, mode=mode, line_capacity=512, want_rois=True)
        return out

    run(warmup)
    # Timed region: ONE hipEvent pair per step around the 13 conv3x3 launches (the roofline kernel) on the ctx stream; a pair
    # around every stage (42 records per step) costs ~0.2 ms of bubbles per step, so the per-stage split is taken in a
    # separate, untimed pass afterwards.
    ctx.profile_enable(2 if stage_events in ("after", "conv_only") else (True if stage_events == "inline" else False))
    ctx.profile_reset()
    sync()
    t0 = time.perf_counter()
    lines = run(steps)
    torch.cuda.synchronize()
    elapsed_local = time.perf_counter() - t0
    sync()
    prof = ctx.profile_read()
    ctx.profile_enable(False)
    stage_steps = steps
    prof_stage = prof
    if stage_events == "after":
        ctx.profile_enable(True)
        ctx.profile_reset()
        stage_steps = min(steps, 5)
        run(stage_steps)
        torch.cuda.synchronize()
        prof_stage = ctx.profile_read()
        ctx.profile_enable(False)
    return elapsed_local, prof, prof_stage, stage_steps, lines


def timed_batch_equals_sample(lines, rois, sample):
    """The LAST TIMED step's output checked against something: images 0 .. k-1 of the timed batch are the images of the accuracy sample
    (same seeds), whose rois and text lines came from a separate k-image synchronous ctx of the same precision and were compared with the
    oracle (`accuracy`). A batch equals its images alone bit for bit (tests/test_gpu_parity.py::test_batch_equals_singles_and_is_idempotent,
    tests/test_gpu_round6.py at n = 32), so the two must be IDENTICAL arrays: a kernel that skipped or mis-tiled work at the timed batch
    size cannot print a number with this field true. sample = (rois list, lines list)."""
    s_rois, s_lines = sample
    k = min(len(s_rois), len(rois))
    return bool(k > 0 and all(np.array_equal(rois[i], s_rois[i]) and np.array_equal(lines[i], s_lines[i]) for i in range(k)))


def secondary_config(ctpn_amd, torch, dev, arena, precision, B, H, W, mode, steps, warmup, host=False, pinned=True, options=None, sample=None):
    """One of the other_configs: its own ctx, timed like the headline (N = 1: no barrier), reported compactly. sample: (rois, lines) of the
    accuracy sample at this precision and geometry -> the field timed_batch_equals_sample."""
    imgs = torch.from_numpy(np.stack([np.random.default_rng(1 + i).integers(0, 256, size=(H, W, 3), dtype=np.uint8) for i in range(B)])).to(dev)
    torch.cuda.synchronize()
    host_images = None
    if host:
        ht = imgs.cpu()
        host_images = (ht.pin_memory() if pinned else ht).numpy()
    with ctpn_amd.Context(dev.index or 0, B, H, W, precision, options=options) as ctx:
        ctx.load_weights(arena)
        el, prof, _, _, (lines, rois) = run_config(ctpn_amd, torch, dev, ctx, imgs, (B, H, W), steps, warmup, mode, host_images=host_images,
                                                   stage_events="conv_only", sync=torch.cuda.synchronize)
    cg = prof["conv_gemm"]
    tf = cg["work"] / (cg["ms"] * 1e-3) / 1e12 if cg["ms"] > 0 else 0.0
    out = {"workload": "batch=%d at %dx%d, %s conv stack, DETECT_MODE=%s%s" % (B, H, W, precision, mode, (", host-resident uint8 images (%s), H2D copy inside the timed region" % ("page-locked" if pinned else "pageable")) if host else ""),
           "images_per_s": round(B * steps / el, 2), "ms_per_step": round(el / steps * 1e3, 3), "steps": steps, "warmup": warmup,
           "conv_stack_tflops": round(tf, 2), "conv_stack_frac_of_peak": round(tf / PEAK[precision], 4), "dtype": precision,
           "lines_last_step": int(sum(len(l) for l in lines))}
    if sample is not None:
        out["timed_batch_equals_sample"] = timed_batch_equals_sample(lines, rois, sample)
    if MFMA_PER_PRODUCT[precision] > 1:      # algorithmic flops above; what the matrix cores actually issue
        out["conv_stack_issued_mfma_tflops"] = round(tf * MFMA_PER_PRODUCT[precision], 2)
        out["conv_stack_issued_frac_of_peak"] = round(tf * MFMA_PER_PRODUCT[precision] / PEAK[precision], 4)
    return out


def single_image_latency(ctpn_amd, torch, dev, arena, precision, H, W, mode, n=100, warm=10, options=None):
    """The reference's calling convention (ctpn/demo.py:55-68): ONE image per call, synchronous, nothing in flight. Median / p10 / p90 of
    n calls each of (a) ctpn_detect on an image resident in HBM -- submit -> collect of a lone image: forward, proposal layer, connector
    front end, D2H, host connector --, (b) the same from a host image (H2D inside), and (c) the drop-in pair test_ctpn(sess, net, im) +
    TextDetector().detect(...) exactly as demo.py:61-64 calls them (host image in, Python seams included)."""
    img_host = np.random.default_rng(1).integers(0, 256, size=(1, H, W, 3), dtype=np.uint8)
    img_dev = torch.from_numpy(img_host).to(dev)
    torch.cuda.synchronize()

    def stats(ts):
        ts = np.sort(np.asarray(ts)) * 1e3
        return {"median_ms": round(float(np.median(ts)), 4), "p10_ms": round(float(ts[len(ts) // 10]), 4), "p90_ms": round(float(ts[(len(ts) * 9) // 10]), 4)}

    def timed(f):
        for _ in range(warm):
            f()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            f()
            ts.append(time.perf_counter() - t0)
        return stats(ts)
    out = {"workload": "1 image %dx%d per call, synchronous (nothing in flight), %s conv stack, DETECT_MODE=%s; median of %d calls after %d warm-up calls"
                       % (H, W, precision, mode, n, warm)}
    with ctpn_amd.Context(dev.index or 0, 1, H, W, precision, options=options) as ctx:
        ctx.load_weights(arena)
        out["ctpn_detect_hbm_resident"] = timed(lambda: ctx.detect(device_ptr=img_dev.data_ptr(), shape=(1, H, W), mode=mode, line_capacity=512))
        out["ctpn_detect_host_image"] = timed(lambda: ctx.detect(img_host, mode=mode, line_capacity=512))
    from ctpn_amd.lib.fast_rcnn.config import cfg
    from ctpn_amd.lib.fast_rcnn.test import test_ctpn
    from ctpn_amd.lib.networks.factory import get_network
    from ctpn_amd.lib.text_connector.detectors import TextDetector
    old = (cfg.TEST.PRECISION, cfg.TEST.DETECT_MODE)
    cfg.TEST.PRECISION, cfg.TEST.DETECT_MODE = precision, mode
    try:
        net = get_network("VGGnet_test")
        net.load_arena(arena)
        det = TextDetector()

        def dropin():
            scores, boxes = test_ctpn(None, net, img_host[0])
            return det.detect(boxes, scores[:, np.newaxis], (H, W))
        out["test_ctpn_plus_TextDetector"] = timed(dropin)
        net.close()
    finally:
        cfg.TEST.PRECISION, cfg.TEST.DETECT_MODE = old
    return out


class GpuSampler:
    """Shader clock and package power of device `index` sampled in the background while a run is in flight: hwmon / sysfs where the box
    exposes them, else `rocm-smi --showpower --showclocks` (what tools/r3_power.sh used in round 3). Reported, never required."""

    def __init__(self, index=0, period=0.4):
        import glob
        import threading
        self.period, self.index = period, index
        self.sclk, self.power, self.how = [], [], None
        self._stop = threading.Event()
        self._t = threading.Thread(target=self._run, daemon=True)
        # every card with a hwmon directory is sampled; summary() reports the one with the HIGHEST mean power -- the device under load. (A
        # container that sees ONE GPU still sees every card of the host in sysfs, and card order is not HIP's device order: round 4's first
        # profile run read an idle neighbour, 2396 MHz at 447 W.)
        self._hw = []
        self._own = None        # hwmon directory of THE device this process computes on, found by its PCI bus id (round 6: on a shared host the
                                # card drawing the most power can be a neighbour's -- a profile run read 1547 MHz at 1400 W off another tenant's GPU)
        bus = self._pci_bus_id(index)
        for dev in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
            hm = glob.glob(os.path.join(dev, "hwmon", "hwmon*"))[:1]
            self._hw += hm
            if hm and bus and os.path.basename(os.path.realpath(dev)).lower() == bus:
                self._own = hm[0]
        self._per = {h: ([], []) for h in self._hw}

    @staticmethod
    def _pci_bus_id(index):
        """'0000:05:00.0' of HIP device `index` (hipDeviceGetPCIBusId through ctypes), or None."""
        try:
            import ctypes
            for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6", "/opt/rocm/lib/libamdhip64.so"):
                try:
                    hip = ctypes.CDLL(name)
                    break
                except OSError:
                    hip = None
            if hip is None:
                return None
            buf = ctypes.create_string_buffer(64)
            if hip.hipDeviceGetPCIBusId(buf, 64, int(index)) != 0:
                return None
            return buf.value.decode().lower() or None
        except Exception:
            return None

    def _sysfs(self):
        got = False
        for h in self._hw:
            p = c = None
            for f in ("power1_average", "power1_input"):
                q = os.path.join(h, f)
                if os.path.exists(q):
                    p = int(open(q).read()) / 1e6
                    break
            q = os.path.join(h, "freq1_input")
            if os.path.exists(q):
                c = int(open(q).read()) / 1e6
            if c is not None and p is not None:
                self._per[h][0].append(c)
                self._per[h][1].append(p)
                got = True
        return got

    def _smi(self):
        import re
        import subprocess
        out = subprocess.run(["rocm-smi", "-d", str(self.index), "--showpower", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
        c = re.search(r"sclk clock level:.*?\((\d+)Mhz\)", out)
        p = re.search(r"Power \(W\):\s*([0-9.]+)", out)
        return (float(c.group(1)) if c else None), (float(p.group(1)) if p else None)

    def _run(self):
        while not self._stop.is_set():
            ok = False
            try:
                ok = self._sysfs()
                if ok:
                    self.how = "sysfs hwmon (freq1_input, power1_average) of the card drawing the most power, %d cards sampled" % len(self._hw)
            except Exception:
                pass
            if not ok:
                try:
                    c, p = self._smi()
                    self.how = self.how or "rocm-smi --showpower --showclocks"
                    if c is not None:
                        self.sclk.append(c)
                    if p is not None:
                        self.power.append(p)
                except Exception:
                    pass
            self._stop.wait(self.period)

    def __enter__(self):
        self._t.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        self._t.join(timeout=15)

    def summary(self):
        def mid(v):       # drop the ramp-up / ramp-down samples at both ends
            v = v[len(v) // 5: len(v) - len(v) // 5] if len(v) >= 5 else v
            return round(float(np.mean(v)), 1) if v else None
        busy = [h for h in self._hw if self._per[h][1]]
        if busy:
            if self._own in busy:
                h = self._own
                self.how = "sysfs hwmon (freq1_input, power1_average) of this process's device (matched by PCI bus id), %d cards on the host" % len(self._hw)
            else:
                h = max(busy, key=lambda k: float(np.mean(self._per[k][1])))
            self.sclk, self.power = self._per[h]
        return {"sclk_mhz_mean": mid(self.sclk), "package_power_w_mean": mid(self.power), "samples": len(self.sclk), "source": self.how}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100, help="timed steps (default 100: 0.9 s at 8.8 ms per step; the first steps of a cold part run at lower clocks, see other_configs.sustained_600_steps)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU per step")
    ap.add_argument("--height", type=int, default=600)
    ap.add_argument("--width", type=int, default=900)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp16", "split", "fp32"],
                    help="ctpn_create precision: bf16 (BASELINE.json's dtype; the headline), fp16 (same MFMA rate, 3 more mantissa bits), split ((hi, lo) "
                         "bf16 pairs, three MFMAs per product: parity-grade), fp32 (exact-fp32 MFMA: the correctness gate)")
    ap.add_argument("--mode", default="H", choices=["H", "O"])
    ap.add_argument("--cpu-images", type=int, default=6, help="images in the CPU-baseline / accuracy sample (0 disables both)")
    ap.add_argument("--host-images", action="store_true",
                    help="feed host uint8 batches instead of HBM-resident ones (PCIe-inclusive rate; never the headline value)")
    ap.add_argument("--stage-events", default="after", choices=["after", "inline", "off"],
                    help="per-stage hipEvent pairs: in an extra untimed pass after the timed region (default), inside it, or not at all")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=INT",
                    help="per-ctx option of the C ABI for the headline ctx (ctpn_set_option), e.g. --option lstm_split=0; repeatable")
    ap.add_argument("--lstm-exact", action="store_true",
                    help="BiLSTM recurrent product on exact-fp32 MFMAs (v_mfma_f32_16x16x4_f32) instead of the 16-bit modes' default, three split-bf16 "
                         "terms per product (fp32 state / gates / accumulation, |d| < 3e-5 against the exact kernel)")
    ap.add_argument("--pageable", action="store_true", help="with --host-images: a pageable host buffer (the ctx stages it through its own "
                                                             "page-locked buffer) instead of the default page-locked one")
    ap.add_argument("--zero-data", action="store_true",
                    help="DIAGNOSTIC, not a benchmark: all-zero weights and images (every MFMA operand is zero). The kernels execute the same "
                         "instructions in the same cycles; what changes is the power they draw and with it the clock (tools/r3_clock.sh)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the other_configs runs after the headline (N = 1 runs them by default)")
    ap.add_argument("--latency-only", action="store_true", help="print only the single-image synchronous latency object (bf16, then with --option's options) and exit")
    ap.add_argument("--rccl-timeout", type=int, default=60, help="N > 1: seconds the RCCL weight broadcast may take before this rank gives it up and falls back to gloo")
    ap.add_argument("--weights-via", default="rccl", choices=["rccl", "gloo"],
                    help="N > 1: how the weight arena reaches the other ranks: rccl = ctpn_broadcast_weights_rank (C ABI, RCCL over xGMI; "
                         "default), gloo = a host broadcast through torch.distributed (what the single-GPU two-rank test uses: RCCL refuses "
                         "two ranks on one device)")
    ap.add_argument("--all-ranks-device", type=int, default=None, metavar="D",
                    help="TESTING: every rank uses device D (exercises the N > 1 code path on a one-GPU box); implies --weights-via gloo")
    ap.add_argument("--try-rccl-on-shared-device", action="store_true",
                    help="TESTING, with --all-ranks-device: attempt the RCCL broadcast anyway (RCCL rejects two ranks on one GPU), to exercise "
                         "the loud fallback to the gloo host broadcast")
    ap.add_argument("--baseline-value", type=float, default=None, metavar="IMAGES_PER_S",
                    help="N > 1: the N = 1 images/s of the same box; the line then carries weak_scaling_efficiency = value / (N x this)")
    ap.add_argument("--print-launch", action="store_true",
                    help="TESTING: every rank prints 'rank R/W local L' as it sees the launch and exits (no GPU needed): checks the self-launch of --gpus N")
    default_pmc = next((p for p in (os.path.join(ROOT, "profiles", f) for f in ("r06_pmc.json", "r05_pmc.json", "r04_pmc.json", "r03_pmc.json", "r02_pmc.json", "r01_pmc.json")) if os.path.exists(p)), None)
    ap.add_argument("--traffic-json", default=default_pmc,
                    help="PMC summary (tools/pmc_summary.py over separate rocprofv3 --pmc passes) that fills roofline.traffic")
    args = ap.parse_args()

    # --gpus N with no launcher around us: start the N ranks ourselves (one process per GPU, torch.distributed.run, loopback rendezvous on
    # a free port) -- `python bench.py --gpus 8` and the torchrun form of the docstring are the same measurement. Under a launcher
    # (WORLD_SIZE set) the world size must be the one asked for: a 1-rank line must never be labelled as anything else.
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        sys.stdout.flush()
        os.execv(sys.executable, cmd)
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    if env_world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE); they must agree" % (args.gpus, env_world))

    if args.print_launch:
        print("bench.py launch: rank %s/%s local %s master %s:%s" % (os.environ.get("RANK", "0"), env_world, os.environ.get("LOCAL_RANK", "0"),
                                                                    os.environ.get("MASTER_ADDR", "-"), os.environ.get("MASTER_PORT", "-")), flush=True)
        return

    import torch
    import ctpn_amd
    from ctpn_amd import dist as D
    from ctpn_amd import _binding as BND

    ctx_options = {"lstm_split": 0} if args.lstm_exact else {}
    for kv in args.option:
        k, v = kv.split("=")
        ctx_options[k] = int(v)
    if args.latency_only:
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(0)
        arena = ctpn_amd.make_synthetic_arena(0)
        print(json.dumps({"single_image_sync_latency": single_image_latency(ctpn_amd, torch, dev, arena, args.precision, args.height, args.width, args.mode,
                                                                             options=ctx_options), "ctx_options": ctx_options}), flush=True)
        return
    rank, local_rank, world = D.env_world()
    if world > 1:
        D.init_process_group("gloo")          # rendezvous, barrier and scalar reductions only: the weights travel over RCCL below
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    dev_index = local_rank if args.all_ranks_device is None else args.all_ranks_device
    if args.all_ranks_device is None and (world > torch.cuda.device_count() or local_rank >= torch.cuda.device_count()):
        raise SystemExit("bench.py rank %d: LOCAL_RANK %d needs its own device, but %d device(s) are visible (HIP_VISIBLE_DEVICES / "
                         "ROCR_VISIBLE_DEVICES = %s / %s); one process per GPU -- --all-ranks-device D is the single-GPU test of the N > 1 code path"
                         % (rank, local_rank, torch.cuda.device_count(), os.environ.get("HIP_VISIBLE_DEVICES", "unset"), os.environ.get("ROCR_VISIBLE_DEVICES", "unset")))
    weights_via = "gloo" if (args.all_ranks_device is not None and not args.try_rccl_on_shared_device) else args.weights_via
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    B, H, W = args.batch, args.height, args.width

    # weights: rank 0 builds the arena and loads it; ONE broadcast (RCCL, C ABI) puts it into every other rank's HBM, which packs it
    arena = ctpn_amd.make_synthetic_arena(0) if rank == 0 else None
    if args.zero_data and arena is not None:
        arena = np.zeros_like(arena)
    ctx = ctpn_amd.Context(dev_index, B, H, W, args.precision, options=ctx_options)
    t_b0 = time.time()
    bcast_how = "none (1 rank)"
    rccl_abandoned = False      # a helper thread is still inside RCCL (communicator never formed): this process must not wait for it at exit
    if world == 1:
        ctx.load_weights(arena)
    else:
        if rank == 0:
            ctx.load_weights(arena)
        done = False
        if weights_via == "rccl":
            # watchdog: a communicator that cannot form (a rank missing, a fabric problem) blocks inside RCCL for ever; say so and end
            # this rank instead of holding the node until the driver's own limit
            import threading
            wd = threading.Timer(180.0, lambda: (print("bench.py rank %d: the RCCL weight broadcast did not finish within 180 s; "
                                                       "re-run with --weights-via gloo" % rank, file=sys.stderr, flush=True), os._exit(17)))
            wd.daemon = True
            wd.start()
            try:
                # every rank takes part in the id hand-over even if the root could not create one (an all-zero id says so): the ranks
                # must stay in step on the side channel whatever RCCL does
                uid_local, err = None, None
                if rank == 0:
                    try:
                        uid_local = BND.comm_unique_id()
                    except ctpn_amd.CtpnError as e:
                        uid_local, err = bytes(BND.COMM_ID_BYTES), e
                uid = D.broadcast_bytes(uid_local, BND.COMM_ID_BYTES, src=0)
                if any(uid):
                    # the collective runs on a helper thread with its own, shorter limit: ctpn_broadcast_weights_rank has no timeout (it
                    # blocks inside ncclCommInitRank if a peer never joins), and a hang here must cost this run the RCCL path, not its result
                    box = {}

                    def _rccl():
                        try:
                            ctx.broadcast_weights_rank(uid, rank, world, root=0)
                            box["ok"] = True
                        except ctpn_amd.CtpnError as e:
                            box["err"] = e
                    th = threading.Thread(target=_rccl, daemon=True)
                    th.start()
                    th.join(args.rccl_timeout)
                    if th.is_alive():
                        # still inside RCCL: abandon that ctx (never touched again) and carry on with a fresh one over the gloo fallback
                        err = "no completion within %d s" % args.rccl_timeout
                        rccl_abandoned = True
                        ctx = ctpn_amd.Context(dev_index, B, H, W, args.precision, options=ctx_options)
                        if rank == 0:
                            ctx.load_weights(arena)
                    elif box.get("ok"):
                        done = True
                        bcast_how = "ctpn_broadcast_weights_rank (RCCL ncclBroadcast of the 71.57 MB fp32 arena on the ctx stream)"
                    else:
                        err = box.get("err")
                if not done:       # loud, and recorded in the JSON line
                    print("bench.py rank %d: RCCL broadcast through the C ABI failed (%s); falling back to a host broadcast over gloo" % (rank, err), file=sys.stderr, flush=True)
                    bcast_how = "gloo host broadcast (C-ABI RCCL path failed: %s)" % str(err)[:120]
            finally:
                wd.cancel()
        # every rank must take the same branch: the fallback runs if ANY rank failed
        if D.min_over_ranks(1.0 if done else 0.0) < 0.5:
            host = D.broadcast_arena(arena, "cpu", src=0).numpy()
            if rank != 0:
                ctx.load_weights(host)
            if weights_via == "gloo":
                bcast_how = "gloo host broadcast (--weights-via gloo)"
            elif done:
                bcast_how = "gloo host broadcast (the C-ABI RCCL path failed on another rank)"
    torch.cuda.synchronize()
    t_bcast = time.time() - t_b0

    # this rank's shard of the global image list (seeds 1 .. world*B), resident in HBM before the timed region
    lo, hi = D.shard_range(world * B, rank, world)
    imgs = torch.from_numpy(np.stack([np.random.default_rng(1 + i).integers(0, 256, size=(H, W, 3), dtype=np.uint8)
                                      for i in range(lo, hi)])).to(dev)
    if args.zero_data:
        imgs.zero_()
    torch.cuda.synchronize()
    shape = (hi - lo, H, W)
    imgs_host = None
    if args.host_images:
        ht = imgs.cpu()
        imgs_host = (ht if args.pageable else ht.pin_memory()).numpy()

    def sync():
        D.barrier()
        torch.cuda.synchronize()
        D.barrier()

    # shader clock and package power over the headline run (rank 0's device; 50 ms period: a sysfs read per card): roofline.sclk_mhz_mean
    with GpuSampler(dev_index, period=0.05) as head_smp:
        elapsed_local, prof, prof_stage, stage_steps, (lines, last_rois) = run_config(ctpn_amd, torch, dev, ctx, imgs, shape, args.steps, args.warmup, args.mode,
                                                                                       host_images=imgs_host, stage_events=args.stage_events, sync=sync)
    head_clock = head_smp.summary()
    elapsed = D.max_over_ranks(elapsed_local, "cpu")
    per_rank = D.gather_over_ranks([elapsed_local / args.steps * 1e3, t_bcast * 1e3, ctx.host_threads()], "cpu")
    fused1 = args.precision in ("bf16", "fp16") and ctx.get_option("conv1_kernel") == 2 and ctx.get_option("conv1_fuse") == 1 and ctx.get_option("keep_acts") == 0
    ctx.close()

    if rank == 0:
        total_images = world * B * args.steps
        cg = prof["conv_gemm"]
        achieved = cg["work"] / (cg["ms"] * 1e-3) / 1e12 if cg["ms"] > 0 else 0.0
        # issued / algorithmic MFMA flops of the family: the precision's MFMAs per product, and -- conv1_1 inside conv1_2 -- its 72 MFMAs of
        # 32768 flops per 8 x 32-pixel tile of conv1_2 (pooled extent) against its 2 x 27 x 64 algorithmic flops per pixel
        issued_ratio = float(MFMA_PER_PRODUCT[args.precision])
        if fused1 and cg["work"] > 0:
            tiles = (((W & ~1) + 31) // 32) * (((H & ~1) + 7) // 8)
            c11_alg, c11_iss = 2.0 * 27 * 64 * H * W, tiles * 72 * 32768.0
            per_img = cg["work"] / max(cg["launches"] / 13.0, 1.0) / B          # algorithmic flops of the family per image (conv1_1 included)
            issued_ratio = (per_img - c11_alg + c11_iss) / per_img
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
            "dtype": {"split": "bf16 pairs (CTPN_PREC_SPLIT: (hi, lo) bf16 per value, three bf16 MFMAs per product, fp32 accumulate)"}.get(args.precision, args.precision), "data": ("ALL-ZERO weights and images: a clock diagnostic, NOT a benchmark" if args.zero_data else "synthetic") + ((" (host-resident, %s, H2D copy inside the timed region)" % ("pageable" if args.pageable else "page-locked")) if args.host_images else ""),
            "config": {"workload": ("batch=%d at %dx%d per GPU, %s MFMA conv stack + fp32 BiLSTM (%s) + HIP proposal/NMS + text lines (%s); "
                                    "BASELINE.json configs[2], sharded as configs[3] for N>1") % (
                                        B, H, W, args.precision,
                                        "exact-fp32 MFMA recurrence" if (args.lstm_exact or args.precision == "fp32" or ctx_options.get("lstm_split") == 0) else
                                        "fp32 state, gates and accumulation; recurrent product as three split-bf16 MFMA terms, within 3e-5 of the exact-fp32 kernel", args.mode),
                       "images_per_gpu": B, "global_batch": world * B, "height": H, "width": W,
                       "parallelism": ("1 rank: weights loaded from the host (%.1f ms), no collective" % (t_bcast * 1e3)) if world == 1 else
                                      ("data-parallel replicas, %d ranks (one process per GPU), one weight broadcast + pack (%.1f ms), no per-batch collective" % (world, t_bcast * 1e3)),
                       "weight_broadcast": bcast_how,
                       "weights": "seeded random init (ctpn_amd.make_synthetic_arena(0)); no trained checkpoint exists in the reference tree",
                       "lines_rank0_last_step": int(sum(len(l) for l in lines)),
                       "host_threads_per_rank": int(per_rank[0][2]), "ctx_options": ctx_options},
            "per_rank": {"ms_per_step": [round(r[0], 3) for r in per_rank], "weight_broadcast_ms": [round(r[1], 1) for r in per_rank]},
            "roofline": {"kernel": "ctpn::conv3x3_wr_kernel x2 (conv1_2%s, conv2_1: weights in registers) + ctpn::conv3x3_p_kernel x11 (tap-reuse MFMA conv3x3 + bias + ReLU "
                                   "(+ 2x2 max-pool)), 13 launches per step (+ conv3x3_edge_kernel launches for ragged tile columns, concurrent with their layers); one hipEvent pair per step around them, gaps included"
                                   % (" with conv1_1 computed in its window stage" if fused1 else ""), "bound": "mfma",
                         "achieved": round(achieved, 2), "peak": PEAK[args.precision], "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK[args.precision], 4), "traffic": traffic,
                         "traffic_source": (os.path.basename(args.traffic_json) + " (separate rocprofv3 --pmc passes of this command; not measured in this run)") if traffic is not None else None,
                         "launches": cg["launches"], "avg_launch_ms": round(cg["ms"] / max(cg["launches"], 1), 4),
                         "flops_per_launch_avg": cg["work"] / max(cg["launches"], 1),
                         "flops_per_image": (CONV_GFLOP_PER_IMAGE_600x900 + (CONV1_1_GFLOP_PER_IMAGE_600x900 if fused1 else 0.0)) * 1e9 if (H, W) == (600, 900) else None,
                         "conv1_1_in_family": bool(fused1),
                         "achieved_is": "ALGORITHMIC flops (2 x MACs of the %s) / time; issued MFMA flops = achieved x %.4f%s" % (
                             "14 layers: conv1_1 runs inside conv1_2's launch" if fused1 else "13 layers", issued_ratio,
                             "; conv1_1 is issued as 72 MFMAs per 8 x 32-pixel tile (window halo, K 27 -> 48): 5.13 GFLOP per 600 x 900 image for its 1.87 algorithmic ones" if fused1 else ""),
                         "issued_mfma_tflops": round(achieved * issued_ratio, 2),
                         # the part runs this load at its package power cap, below the 2400 MHz the peak is quoted at (MI355X_MICROARCH.md:33;
                         # DESIGN section 4, profiles/r06_gemm_yardstick.txt): the clock over THIS run (warm-up, timed region and the stage pass) and
                         # the fraction of the peak at that clock
                         "sclk_mhz_mean": head_clock["sclk_mhz_mean"], "package_power_w_mean": head_clock["package_power_w_mean"],
                         "clock_samples": head_clock["samples"], "clock_source": head_clock["source"],
                         "frac_of_clock_adjusted_peak": (round(achieved / (PEAK[args.precision] * head_clock["sclk_mhz_mean"] / 2400.0), 4)
                                                         if head_clock["sclk_mhz_mean"] else None)},
            "stages_ms_per_step": {k: round((prof_stage["conv_gemm"]["ms"] if k == "conv_gemm" else v["ms"]) / stage_steps, 4) for k, v in prof_stage.items()},
            "stage_events": args.stage_events,
        }
        if world > 1 and args.baseline_value:
            out["weak_scaling_efficiency"] = round(out["value"] / (world * args.baseline_value), 4)
            out["weak_scaling_baseline_images_per_s"] = args.baseline_value
        if world == 1:
            oracle_out = None
            if args.cpu_images > 0:
                out["cpu_baseline"], oracle_out = cpu_baseline(arena, H, W, args.cpu_images, args.mode)
                cls, rois, dlines = device_sample_outputs(ctpn_amd, arena, args.precision, H, W, len(oracle_out), args.mode)
                out["accuracy"] = accuracy_against(oracle_out, cls, rois, dlines, geom=(H, W, args.mode))
                out["accuracy"]["path"] = "%s conv stack (this run's configuration)" % args.precision
                if not ctx_options and not args.zero_data:
                    # the timed batch's own output (last timed step, images 0 .. k-1) against the sample that `accuracy` judged
                    out["timed_batch_equals_sample"] = timed_batch_equals_sample(lines, last_rois, (rois, dlines))
            if not args.no_other_configs and (B, H, W, args.precision) == (32, 600, 900, "bf16") and not args.host_images:
                oc = {}
                # the accuracy samples first (each precision's 6-image synchronous ctx against the oracle run of the cpu_baseline leg), so that
                # every timed batch below can be checked against the sample of its precision (timed_batch_equals_sample)
                samples, acc = {}, {}
                if oracle_out is not None:
                    for prec in ("fp32", "split", "fp16"):
                        cls, rois, dlines = device_sample_outputs(ctpn_amd, arena, prec, H, W, len(oracle_out), args.mode)
                        acc[prec] = accuracy_against(oracle_out, cls, rois, dlines, geom=(H, W, args.mode))
                        samples[prec] = (rois, dlines)
                hires_ref, hires_acc, hires_sample = None, {}, None
                if oracle_out is not None:
                    # config 5's geometry end to end against the oracle (2 images: the oracle forward at 1280 x 1920 is ~7 s each), the
                    # bench's own mode and the two parity-grade ones
                    hires_ref = oracle_outputs(arena, 1280, 1920, 2, "O")
                    for prec in ("bf16", "split", "fp32"):
                        cls, rois, dlines = device_sample_outputs(ctpn_amd, arena, prec, 1280, 1920, len(hires_ref), "O")
                        hires_acc[prec] = accuracy_against(hires_ref, cls, rois, dlines, geom=(1280, 1920, "O"))
                        if prec == "bf16":
                            hires_sample = (rois, dlines)
                oc["config5_hires_O"] = secondary_config(ctpn_amd, torch, dev, arena, "bf16", 8, 1280, 1920, "O", 8, 2, sample=hires_sample)
                oc["fp32_gate_b8"] = secondary_config(ctpn_amd, torch, dev, arena, "fp32", 8, 600, 900, args.mode, 8, 2, sample=samples.get("fp32"))
                oc["fp32_gate_b32"] = secondary_config(ctpn_amd, torch, dev, arena, "fp32", 32, 600, 900, args.mode, 4, 1, sample=samples.get("fp32"))
                oc["split_b32"] = secondary_config(ctpn_amd, torch, dev, arena, "split", 32, 600, 900, args.mode, 6, 2, sample=samples.get("split"))
                oc["fp16_b32"] = secondary_config(ctpn_amd, torch, dev, arena, "fp16", 32, 600, 900, args.mode, 20, 3, sample=samples.get("fp16"))
                oc["split_b32"]["speedup_vs_fp32_gate_b32"] = round(oc["split_b32"]["images_per_s"] / oc["fp32_gate_b32"]["images_per_s"], 3)
                if oracle_out is not None:
                    for key, prec in (("fp32_gate_b8", "fp32"), ("split_b32", "split"), ("fp16_b32", "fp16")):
                        oc[key]["accuracy"] = acc[prec]
                    oc["fp32_gate_b32"]["accuracy"] = "see fp32_gate_b8 (same kernels, results do not depend on the batch: timed_batch_equals_sample)"
                    oc["config5_hires_O"]["accuracy"] = hires_acc
                # what `python ctpn/demo.py` and test_ctpn run out of the box is NOT the headline's arithmetic: cfg.TEST.PRECISION /
                # ctpn/text.yml default to split, the parity-grade mode; its rate, next to the headline, at the top level of the line
                out["drop_in_default_precision"] = {"precision": "split (cfg.TEST.PRECISION, ctpn/text.yml)", "images_per_s": oc["split_b32"]["images_per_s"],
                                                    "ms_per_step": oc["split_b32"]["ms_per_step"], "headline_precision": args.precision,
                                                    "note": "the headline `value` is BASELINE.json configs[2]'s dtype (bf16), which is outside north_star's tolerance; the drop-in's "
                                                            "default is the parity-grade split mode: other_configs.split_b32 (with its accuracy), single_image_sync_latency_split"}
                oc["bf16_exact_fp32_recurrence_b32"] = secondary_config(ctpn_amd, torch, dev, arena, "bf16", 32, 600, 900, args.mode, 20, 3, options={"lstm_split": 0})
                oc["batch1_pipelined"] = secondary_config(ctpn_amd, torch, dev, arena, "bf16", 1, 600, 900, args.mode, 100, 10)
                oc["batch1_pipelined"]["note"] = "THROUGHPUT at batch 1 with two submits in flight (ms_per_step = time per image), not a latency: see single_image_sync_latency"
                oc["single_image_sync_latency"] = single_image_latency(ctpn_amd, torch, dev, arena, "bf16", 600, 900, args.mode)
                oc["single_image_sync_latency_split"] = single_image_latency(ctpn_amd, torch, dev, arena, "split", 600, 900, args.mode, n=50, warm=5)
                oc["host_images_pcie_inclusive"] = secondary_config(ctpn_amd, torch, dev, arena, "bf16", 32, 600, 900, args.mode, 10, 3, host=True, pinned=True)
                oc["host_images_pcie_inclusive_pageable"] = secondary_config(ctpn_amd, torch, dev, arena, "bf16", 32, 600, 900, args.mode, 10, 3, host=True, pinned=False)
                with GpuSampler(dev.index or 0) as smp:
                    oc["sustained_600_steps"] = secondary_config(ctpn_amd, torch, dev, arena, "bf16", 32, 600, 900, args.mode, 600, 20)
                oc["sustained_600_steps"].update(smp.summary())
                oc["sustained_600_steps"]["vs_headline_value"] = round(oc["sustained_600_steps"]["images_per_s"] / out["value"], 4)
                oc["note"] = "run after the headline's timed region, each on its own ctx, same timing discipline (warm-up, then exactly `steps` fully collected passes); never `value`"
                out["other_configs"] = oc
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        D.barrier()
        dist.destroy_process_group()
    if rccl_abandoned:
        # the abandoned helper thread sits inside ncclCommInitRank and holds RCCL / HIP state that the runtime's exit handlers would wait for:
        # the line is printed and flushed, every rank has passed the last barrier -- leave without running them
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)


if __name__ == "__main__":
    main()
