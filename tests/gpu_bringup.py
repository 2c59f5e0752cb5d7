"""GPU bring-up report (a script, not a pytest module): layer-by-layer comparison of the HIP path against the
oracle, for every kernel variant, with enough detail in gpurun_out/bringup.json to debug offline.

    python tests/gpu_bringup.py [--quick] [--out gpurun_out/bringup.json]
"""
import argparse
import json
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

REPORT = {"sections": {}}


def section(name):
    def deco(fn):
        def run(*a, **k):
            t0 = time.time()
            try:
                res = fn(*a, **k)
                REPORT["sections"][name] = {"ok": True, "result": res, "sec": round(time.time() - t0, 2)}
            except Exception as e:  # noqa: BLE001
                REPORT["sections"][name] = {"ok": False, "error": repr(e), "trace": traceback.format_exc(), "sec": round(time.time() - t0, 2)}
            print("[%s] %s" % (name, json.dumps(REPORT["sections"][name], default=str)[:3000]), flush=True)
        return run
    return deco


def err_stats(got, ref):
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    if got.shape != ref.shape:
        return {"shape_mismatch": [list(got.shape), list(ref.shape)]}
    d = np.abs(got - ref)
    scale = max(float(np.abs(ref).max()), 1e-30)
    idx = np.unravel_index(int(np.argmax(d)), d.shape) if d.size else ()
    return {"max_abs": float(d.max()) if d.size else 0.0, "mean_abs": float(d.mean()) if d.size else 0.0,
            "ref_absmax": scale, "rel_max": float(d.max() / scale) if d.size else 0.0,
            "argmax": [int(i) for i in idx], "nan": int(np.isnan(got).sum()),
            "frac_gt_1e-3rel": float((d > 1e-3 * scale).mean()) if d.size else 0.0}


def canon_rois(r):
    r = np.asarray(r)
    if r.shape[0] == 0:
        return r
    key = np.lexsort((r[:, 4], r[:, 3], r[:, 2], r[:, 1], -r[:, 0].astype(np.float64)))
    return r[key]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "bringup.json"))
    args = ap.parse_args()
    os.makedirs(os.path.dirname(args.out), exist_ok=True)

    import ctpn_amd
    from ctpn_amd import _binding as B
    from oracle import network as N
    from oracle import postproc as P
    from oracle.make_golden import synth_inputs, CASES

    @section("env")
    def env():
        maps = open("/proc/self/maps").read()
        libs = sorted({l.split()[-1] for l in maps.splitlines() if "libamdhip64" in l or "libctpn_hip" in l})
        return {"devices": B.device_count(), "abi": B.load_library().ctpn_abi_version(), "libs": libs, "cpus": os.cpu_count()}
    env()

    arena = ctpn_amd.make_synthetic_arena(0)
    W = ctpn_amd.arena_views(arena)

    def layerwise(prec, variant, n, h, w, seed0):
        os.environ["CTPN_IGEMM_VARIANT"] = str(variant & 1)
        os.environ["CTPN_CONV_IMPL"] = str(variant >> 1)
        os.environ["CTPN_KEEP_ACTS"] = "1"
        imgs = ctpn_amd.weights.synthetic_images(n, h, w, seed0)
        res = {}
        with ctpn_amd.Context(0, n, h, w, prec) as ctx:
            ctx.load_weights(arena)
            ctx.forward(imgs)
            ctx.sync()
            full = N.forward(imgs, W)
            prev_dev = None
            # isolated: oracle layer applied to the DEVICE's previous activation
            x = N.image_blob(imgs)
            for name in N.CONVS:
                dev = ctx.get_tensor(name)
                src = x if prev_dev is None else prev_dev
                iso = N.conv3x3_relu(src, W[name + "/weights"], W[name + "/biases"])
                res[name] = {"iso": err_stats(dev, iso), "cum": err_stats(dev, full[name])}
                prev_dev = dev
                if name in N.POOL_AFTER:
                    pn = N.POOL_AFTER[name]
                    pdev = ctx.get_tensor(pn)
                    res[pn] = {"iso": err_stats(pdev, N.maxpool2x2(dev)), "cum": err_stats(pdev, full[pn])}
                    prev_dev = pdev
            pre = ctx.get_tensor("lstm_pre")
            res["lstm_pre"] = {"iso": err_stats(pre, N.lstm_pre(prev_dev, W)), "cum": err_stats(pre, full["lstm_pre"])}
            lo = ctx.get_tensor("lstm_out")
            res["lstm_out"] = {"iso": err_stats(lo, N.bilstm(prev_dev, W)), "cum": err_stats(lo, full["lstm_out"])}
            fc = ctx.get_tensor("lstm_o")
            res["lstm_o"] = {"iso": err_stats(fc, N.dense(lo, W["lstm_o/weights"], W["lstm_o/biases"])), "cum": err_stats(fc, full["lstm_o"])}
            hd = ctx.get_tensor("heads")
            hb = N.dense(fc, W["rpn_bbox_pred/weights"], W["rpn_bbox_pred/biases"])
            hc = N.dense(fc, W["rpn_cls_score/weights"], W["rpn_cls_score/biases"])
            res["heads"] = {"iso": err_stats(hd, np.concatenate([hb, hc], -1))}
            info = np.array([[h, w, 1.0]] * n, np.float32)
            rois = ctx.proposals(info)
            cp = ctx.get_tensor("rpn_cls_prob_reshape")
            bp = ctx.get_tensor("rpn_bbox_pred")
            res["cls_prob"] = {"iso": err_stats(cp, N.pair_softmax(hd[..., 40:60])), "cum": err_stats(cp, full["rpn_cls_prob_reshape"])}
            res["bbox_pred"] = {"iso": err_stats(bp, hd[..., :40]), "cum": err_stats(bp, full["rpn_bbox_pred"])}
            # proposals: oracle proposal layer on the DEVICE's head outputs
            pr = []
            for i in range(n):
                ref = P.proposal_layer(cp[i:i + 1], bp[i:i + 1], info[i])
                got = rois[i]
                e = {"n_got": int(got.shape[0]), "n_ref": int(ref.shape[0])}
                if got.shape == ref.shape:
                    e["max_abs_sorted"] = float(np.abs(canon_rois(got) - canon_rois(ref)).max()) if got.size else 0.0
                    e["exact_rows"] = int((np.abs(got - ref).max(axis=1) == 0).sum()) if got.size else 0
                pr.append(e)
            res["proposals"] = pr
        summ = {k: (v["iso"].get("rel_max"), v.get("cum", {}).get("rel_max")) for k, v in res.items() if isinstance(v, dict)}
        return {"summary_rel_iso_cum": summ, "detail": res}

    for prec in ("fp32", "bf16"):
        for variant in (3, 1):
            section("layerwise_%s_v%d_small" % (prec, variant))(layerwise)(prec, variant, 2, 70, 100, 101)
    section("layerwise_fp32_v3_mid")(layerwise)("fp32", 3, 1, 300, 452, 7)
    os.environ["CTPN_KEEP_ACTS"] = "0"
    os.environ["CTPN_CONV_IMPL"] = "1"

    @section("proposals_from_golden")
    def props():
        out = {}
        os.environ["CTPN_IGEMM_VARIANT"] = "1"
        with ctpn_amd.Context(0, 1, 608, 1296, "fp32") as ctx:
            for tag, seed, hf, wf, imh, imw in CASES:
                g = np.load(os.path.join(ROOT, "tests", "golden", "postproc_%s.npz" % tag))
                cls, bbox = synth_inputs(seed, hf, wf)
                got = ctx.proposals_from_host(cls, bbox, g["im_info"])[0]
                ref = g["rois"]
                e = {"n_got": int(got.shape[0]), "n_ref": int(ref.shape[0])}
                if got.shape == ref.shape:
                    d = np.abs(canon_rois(got) - canon_rois(ref))
                    e["max_abs_sorted"] = float(d.max())
                    e["rows_exact"] = int((d.max(axis=1) == 0).sum())
                out[tag] = e
                # connector on the golden rois, device NMS
                for mode in "HO":
                    r = B.text_lines(ref[:, 1:5], ref[:, 0], (imh, imw), mode, device_id=0)
                    gr = g["recs_" + mode]
                    out[tag + "_lines_" + mode] = {"n": [int(r.shape[0]), int(gr.shape[0])],
                                                   "max_abs": float(np.abs(r - gr).max()) if r.shape == gr.shape and r.size else None}
                dets = np.hstack([ref[:, 1:5], ref[:, 0:1]]).astype(np.float32)
                keep = B.nms_sorted(dets, 0.2, 0)
                out[tag + "_nms0p2_equal"] = bool(np.array_equal(keep.astype(np.int64), g["nms_keep_0p2"]))
        return out
    props()

    @section("nms_random")
    def nms_rand():
        out = {}
        rng = np.random.default_rng(5)
        for n in (1, 63, 64, 65, 1000, 5000, 12000):
            x1 = rng.uniform(0, 880, n).astype(np.float32)
            y1 = rng.uniform(0, 560, n).astype(np.float32)
            x1 = np.floor(x1 / 16) * 16
            b = np.stack([x1, y1, x1 + 16, y1 + rng.uniform(8, 120, n).astype(np.float32)], 1).astype(np.float32)
            s = np.sort(rng.uniform(0, 1, n).astype(np.float32))[::-1]
            dets = np.hstack([b, s[:, None]]).astype(np.float32)
            for thr in (0.7, 0.2):
                t0 = time.time()
                keep = B.nms_sorted(dets, thr, 0)
                t1 = time.time()
                ref = np.asarray(P.nms(dets, thr))
                out["n%d_t%.1f" % (n, thr)] = {"equal": bool(np.array_equal(keep, ref)), "kept": int(keep.size), "ref": int(ref.size), "ms": round((t1 - t0) * 1e3, 3)}
        return out
    nms_rand()

    if not args.quick:
        @section("full_600x900_fp32")
        def full():
            os.environ["CTPN_IGEMM_VARIANT"] = "1"
            imgs = ctpn_amd.weights.synthetic_images(1, 600, 900, 1)
            t0 = time.time()
            ref = N.forward(imgs, W, keep={"conv5_3", "rpn_conv/3x3", "lstm_o"})
            t_cpu = time.time() - t0
            info = np.array([[600, 900, 1.0]], np.float32)
            out = {"cpu_forward_s": round(t_cpu, 2)}
            with ctpn_amd.Context(0, 1, 600, 900, "fp32") as ctx:
                ctx.load_weights(arena)
                ctx.forward(imgs); ctx.sync()
                t0 = time.time(); ctx.forward(imgs); ctx.sync(); out["gpu_forward_ms"] = round((time.time() - t0) * 1e3, 2)
                rois = ctx.proposals(info)[0]
                for nm in ("conv5_3", "rpn_conv/3x3", "lstm_o"):
                    out[nm] = err_stats(ctx.get_tensor(nm), ref[nm])
                cp = ctx.get_tensor("rpn_cls_prob_reshape"); bp = ctx.get_tensor("rpn_bbox_pred")
                out["cls_prob"] = err_stats(cp, ref["rpn_cls_prob_reshape"])
                out["bbox_pred"] = err_stats(bp, ref["rpn_bbox_pred"])
                fg = cp.reshape(-1, 2)[:, 1]
                out["fg_stats"] = [float(fg.min()), float(fg.mean()), float(fg.max()), float((fg > 0.7).mean())]
                ref_rois = P.proposal_layer(ref["rpn_cls_prob_reshape"], ref["rpn_bbox_pred"], info[0])
                out["rois_n"] = [int(rois.shape[0]), int(ref_rois.shape[0])]
                if rois.shape == ref_rois.shape:
                    out["rois_max_abs_sorted"] = float(np.abs(canon_rois(rois) - canon_rois(ref_rois)).max())
                for mode in "HO":
                    a = B.text_lines(rois[:, 1:5], rois[:, 0], (600, 900), mode, 0)
                    b = P.text_detect(ref_rois[:, 1:5], ref_rois[:, 0], (600, 900), mode)
                    out["lines_" + mode] = {"n": [int(a.shape[0]), int(b.shape[0])], "max_abs": float(np.abs(a - b).max()) if a.shape == b.shape and a.size else None}
                np.savez_compressed(os.path.join(os.path.dirname(args.out), "full_fp32_outputs.npz"), rois=rois, ref_rois=ref_rois)
            with ctpn_amd.Context(0, 1, 600, 900, "bf16") as ctx:
                ctx.load_weights(arena)
                ctx.forward(imgs); ctx.sync()
                t0 = time.time(); ctx.forward(imgs); ctx.sync(); out["gpu_forward_bf16_ms"] = round((time.time() - t0) * 1e3, 2)
                ctx.proposals(info)
                cp = ctx.get_tensor("rpn_cls_prob_reshape")
                out["bf16_cls_prob"] = err_stats(cp, ref["rpn_cls_prob_reshape"])
                out["bf16_conv5_3"] = err_stats(ctx.get_tensor("conv5_3"), ref["conv5_3"])
            return out
        full()

        def timing(prec, n, variant):
            os.environ["CTPN_IGEMM_VARIANT"] = "1"
            os.environ["CTPN_CONV_IMPL"] = str(variant)
            os.environ["CTPN_KEEP_ACTS"] = "0"
            imgs = ctpn_amd.weights.synthetic_images(n, 600, 900, 1)
            info = np.array([[600, 900, 1.0]] * n, np.float32)
            out = {}
            with ctpn_amd.Context(0, n, 600, 900, prec) as ctx:
                ctx.load_weights(arena)
                for _ in range(2):
                    ctx.forward(imgs); ctx.proposals(info)
                t0 = time.time()
                K = 3
                for _ in range(K):
                    ctx.forward(imgs); ctx.sync()
                out["forward_ms"] = round((time.time() - t0) / K * 1e3, 3)
                t0 = time.time()
                for _ in range(K):
                    ctx.proposals(info)
                out["proposals_ms"] = round((time.time() - t0) / K * 1e3, 3)
                t0 = time.time()
                for _ in range(K):
                    ctx.detect(imgs)
                out["detect_ms"] = round((time.time() - t0) / K * 1e3, 3)
                ctx.profile_enable(True); ctx.profile_reset()
                ctx.forward(imgs); ctx.proposals(info)
                out["profile"] = ctx.profile_read()
                ctx.profile_enable(False)
                cg = out["profile"]["conv_gemm"]
                if cg["ms"] > 0:
                    out["conv_gemm_tflops"] = round(cg["work"] / cg["ms"] / 1e9, 2)
            return out
        os.environ["CTPN_C3_PIPE"] = "1"
        section("timing_bf16_n32_c3")(timing)("bf16", 32, 1)
        section("timing_bf16_n32_igemm")(timing)("bf16", 32, 0)
        section("timing_fp32_n4_c3")(timing)("fp32", 4, 1)
        section("timing_bf16_n1_c3")(timing)("bf16", 1, 1)

    with open(args.out, "w") as f:
        json.dump(REPORT, f, indent=1, default=str)
    bad = [k for k, v in REPORT["sections"].items() if not v["ok"]]
    print("FAILED SECTIONS:", bad)


if __name__ == "__main__":
    main()
