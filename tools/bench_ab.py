#!/usr/bin/env python
"""A/B runs of bench.py on ONE box: `python tools/bench_ab.py [--reps R] [--args "..."] ENV=V[,ENV2=V2] ...`

Every positional argument is one variant (comma-separated environment assignments, "-" = no override); variants
are interleaved R times so that clock / thermal drift between boxes and within a run cancels out of the comparison.
Prints one compact row per run: variant, images/s, ms/step, conv TFLOP/s, and the per-stage milliseconds.
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--args", default="--steps 20 --warmup 3 --cpu-images 0")
    ap.add_argument("variants", nargs="+")
    a = ap.parse_args()
    for r in range(a.reps):
        for v in a.variants:
            env = dict(os.environ)
            if v != "-":
                for kv in v.split(","):
                    k, val = kv.split("=", 1)
                    env[k] = val
            p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + a.args.split(), env=env,
                               capture_output=True, text=True)
            line = [l for l in p.stdout.splitlines() if l.startswith("{")]
            if not line:
                print(v, "FAILED", p.stderr[-400:], flush=True)
                continue
            d = json.loads(line[-1])
            s = d["stages_ms_per_step"]
            print("%-28s %8.1f img/s %7.3f ms  conv %7.1f TF | first %.3f gemm %.3f fc %.3f lstm %.3f sort %.3f nms %.3f" % (
                v, d["value"], d["ms_per_step"], d["roofline"]["achieved"], s["conv_first"], s["conv_gemm"], s["gemm"],
                s["bilstm"], s["sort"], s["nms"]), flush=True)


if __name__ == "__main__":
    main()
