import sys, time, os
sys.path.insert(0, '/root/repo')
import numpy as np, torch, ctpn_amd
arena = ctpn_amd.make_synthetic_arena(0)
dev = torch.device('cuda', 0)
img = torch.from_numpy(np.random.default_rng(1).integers(0, 256, size=(1, 600, 900, 3), dtype=np.uint8)).to(dev)
torch.cuda.synchronize()
with ctpn_amd.Context(0, 1, 600, 900, 'bf16') as ctx:
    ctx.load_weights(arena)
    for k in range(20):
        ctx.detect_submit(device_ptr=img.data_ptr(), shape=(1, 600, 900), slot=k & 1)
        if k: ctx.detect_collect((k - 1) & 1, mode='H', line_capacity=512)
    ctx.detect_collect(1, mode='H', line_capacity=512)
    ts, tc = [], []
    t_all = time.perf_counter()
    N = 400
    for k in range(N):
        t0 = time.perf_counter()
        ctx.detect_submit(device_ptr=img.data_ptr(), shape=(1, 600, 900), slot=k & 1)
        t1 = time.perf_counter()
        if k: ctx.detect_collect((k - 1) & 1, mode='H', line_capacity=512)
        t2 = time.perf_counter()
        ts.append(t1 - t0); tc.append(t2 - t1)
    ctx.detect_collect((N - 1) & 1, mode='H', line_capacity=512)
    tot = time.perf_counter() - t_all
    print("per image %.1f us; submit call median %.1f us (p90 %.1f); collect call median %.1f us (p90 %.1f)" % (tot / N * 1e6, np.median(ts) * 1e6, np.percentile(ts, 90) * 1e6, np.median(tc) * 1e6, np.percentile(tc, 90) * 1e6))
