"""GPU (-m gpu), round 4: what VERDICT r3 asked for.

  * `bench.py --gpus 2` launches its two ranks ITSELF (no torchrun wrapper, no WORLD_SIZE in the environment) and prints n_gpus: 2 --
    on a one-GPU box with both ranks on device 0 (--all-ranks-device 0), on a box with two devices over the RCCL broadcast of the C ABI;
  * ctpn_broadcast_weights([ctx0, ctx1]) across two real devices (skipped, not failed, where only one is visible);
  * the precision modes of round 4 are in test_gpu_precision.py.
Nothing here reads /root/reference.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import ctpn_amd
from ctpn_amd import _binding as B

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

two_devices = pytest.mark.skipif(B.device_count() < 2, reason="needs two visible GPUs (the driver's multi-GPU node)")


def _bench_no_launcher(args, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-1500:]            # rank 0 alone prints the JSON line
    return json.loads(lines[0])


def test_bench_gpus_2_self_launch_on_one_device():
    """`python bench.py --gpus 2 ...` as the driver would type it for N = 1, with no launcher: bench.py re-executes itself under
    torch.distributed.run with two ranks. Both ranks share device 0 here, so the arena travels over gloo (RCCL rejects duplicate devices)."""
    d = _bench_no_launcher(["--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "4", "--all-ranks-device", "0", "--cpu-images", "0"])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8 and len(d["per_rank"]["ms_per_step"]) == 2
    assert "2 ranks" in d["config"]["parallelism"] and "gloo" in d["config"]["weight_broadcast"]
    assert d["value"] > 0 and abs(d["value"] - 8 * 3 / (d["ms_per_step"] * 3e-3)) < 0.02 * d["value"]


def test_bench_one_rank_line_says_one_rank():
    d = _bench_no_launcher(["--steps", "2", "--warmup", "1", "--batch", "2", "--cpu-images", "0", "--no-other-configs"])
    assert d["n_gpus"] == 1 and d["config"]["parallelism"].startswith("1 rank") and d["config"]["weight_broadcast"] == "none (1 rank)"


def test_bench_refuses_more_ranks_than_devices():
    if B.device_count() >= 2:
        pytest.skip("one-GPU boxes only")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "1", "--cpu-images", "0"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode != 0 and "device(s) are visible" in (p.stderr + p.stdout)


@two_devices
def test_bench_gpus_2_over_rccl_on_two_devices():
    d = _bench_no_launcher(["--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "4", "--cpu-images", "0", "--weights-via", "rccl"])
    assert d["n_gpus"] == 2 and len(d["per_rank"]["ms_per_step"]) == 2
    assert "ctpn_broadcast_weights_rank (RCCL" in d["config"]["weight_broadcast"], d["config"]["weight_broadcast"]


@two_devices
def test_broadcast_weights_across_two_devices(arena):
    """ctpn_broadcast_weights(handles[], n) -- one process, one ctx per GPU (SURVEY 8b's export list): after the broadcast the second
    device computes the same `heads` bytes as the root."""
    imgs = ctpn_amd.weights.synthetic_images(1, 96, 160, 3)
    with ctpn_amd.Context(0, 1, 96, 160, "bf16") as c0, ctpn_amd.Context(1, 1, 96, 160, "bf16") as c1:
        c0.load_weights(arena)
        B.broadcast_weights([c0, c1])
        c0.forward(imgs)
        c1.forward(imgs)
        assert np.array_equal(c0.get_tensor("heads"), c1.get_tensor("heads"))


@pytest.mark.parametrize("option", ["tail_overlap"])
def test_overlap_options_give_the_default_paths_bytes(arena, option):
    """option tail_overlap = 1 (BiLSTM + heads of batch k on the proposal stream, next to conv1_1 of batch k + 1) had no test
    that pins it. The same sequence of asynchronous submits / collects -- two pipelined batches, a synchronous ctpn_forward in between, a
    GEOMETRY CHANGE, two more pipelined batches -- must give byte-identical rois and text lines with the option on and off."""
    a = ctpn_amd.weights.synthetic_images(3, 150, 230, 11)
    b = ctpn_amd.weights.synthetic_images(2, 96, 160, 12)
    outs = {}
    for opt in (0, 1):
        with ctpn_amd.Context(0, 3, 150, 230, "bf16", options={option: opt}) as ctx:
            assert ctx.get_option(option) == opt
            ctx.load_weights(arena)
            got = []
            ctx.detect_submit(images=a, slot=0)
            ctx.detect_submit(images=a[::-1].copy(), slot=1)
            got.append(ctx.detect_collect(0, want_rois=True))
            got.append(ctx.detect_collect(1, want_rois=True))
            ctx.forward(a)                                          # a synchronous forward between submits (rewrites xp / lstm_out / heads)
            heads = ctx.get_tensor("heads")
            ctx.detect_submit(images=b, slot=0)                     # geometry change: borders re-zeroed, tail of the previous batch waited for
            ctx.detect_submit(images=a, slot=1)
            got.append(ctx.detect_collect(0, want_rois=True))
            got.append(ctx.detect_collect(1, want_rois=True))
            outs[opt] = (got, heads)
    for (l0, r0), (l1, r1) in zip(outs[0][0], outs[1][0]):
        assert len(l0) == len(l1)
        for x, y in zip(l0, l1):
            assert np.array_equal(x, y)
        for x, y in zip(r0, r1):
            assert np.array_equal(x, y)
    assert np.array_equal(outs[0][1], outs[1][1])
    first, last = outs[0][0][0], outs[0][0][3]                       # batch `a` before and after everything else: the same bytes
    for x, y in zip(first[1], last[1]):
        assert np.array_equal(x, y)
