"""Per-layer hashes of N identical forwards in a fresh process (a diagnosis tool: is the FIRST forward of a process different from the later ones?)
   usage: python tools/first_run_check.py [n] [runs]  -- product API only; prints one line per tensor with the hash of every run"""
import sys, os, hashlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctpn_amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
names = ["pool1", "conv2_1", "pool2", "conv3_1", "conv3_2", "pool3", "conv4_1", "conv4_2", "pool4", "conv5_1", "conv5_2", "conv5_3",
         "rpn_conv/3x3", "lstm_pre", "lstm_out", "heads"]
arena = ctpn_amd.make_synthetic_arena(0)
if os.environ.get("DIRTY"):
    # recycle device pages full of garbage (a fresh process gets zeroed pages from the driver, a long-lived one does not)
    import torch
    fill = int(os.environ["DIRTY"])
    blocks = [torch.full((1 << 30,), fill, dtype=torch.uint8, device="cuda:0") for _ in range(12)]
    torch.cuda.synchronize(); del blocks; torch.cuda.empty_cache()
imgs = ctpn_amd.weights.synthetic_images(n, 600, 900, 1)
hs = {k: [] for k in names + ["rois"]}
with ctpn_amd.Context(0, n, 600, 900, "bf16") as ctx:
    ctx.load_weights(arena)
    for r in range(runs):
        lines, rois = ctx.detect(imgs, want_rois=True)
        for k in names:
            try:
                t = ctx.get_tensor(k)
                hs[k].append(hashlib.md5(t.tobytes()).hexdigest()[:8])
            except Exception as e:
                hs[k].append("n/a")
        hs["rois"].append(hashlib.md5(b"".join(x.tobytes() for x in rois)).hexdigest()[:8])
    first = rois
    bad = 0
    for i in range(n):
        ls, rs = ctx.detect(imgs[i:i + 1], want_rois=True)
        if not np.array_equal(rs[0], first[i]):
            bad += 1
for k in names + ["rois"]:
    v = hs[k]
    print("%-14s %s %s" % (k, " ".join(v), "" if len(set(v)) == 1 else "<-- DIFFERS"))
print("singles differing from batch:", bad, "of", n)
