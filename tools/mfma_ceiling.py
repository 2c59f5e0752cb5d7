#!/usr/bin/env python
"""Runs tools/mfma_ceiling.hip's matrix (matrix pipe alone / operands through LDS / + LDS-DMA streaming, random and all-zero data) on this
box, samples shader clock and package power next to every run (bench.GpuSampler: sysfs hwmon of the card under load), then runs ONE
bench.py headline (300 steps, no other configs) under the same sampler -- the conv stack's TFLOP/s, clock and power on the same box, in the
same minute. Prints a table and writes it with the raw JSON lines (VERDICT r5 "next" 3: the power-ceiling argument on file).

    python tools/mfma_ceiling.py [--seconds 4] [--out gpurun_out/r6ceil/mfma_ceiling.txt]
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--out", default=None)
    ap.add_argument("--bench-steps", type=int, default=300)
    args = ap.parse_args()
    from bench import GpuSampler
    exe = "/tmp/mfma_ceiling"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-o", exe, os.path.join(ROOT, "tools", "mfma_ceiling.hip")], check=True)
    rows = []
    # LDS reads per four MFMAs: 4 = one fragment per MFMA; 3.3 ~ the persistent conv kernel (2 x + 2 w reads per 4 MFMAs, a third of the x reads carried);
    # 2 = what a 512-pixel x 128-channel (or 256 x 256) wave tile of 4 x 2 accumulators would need (the "larger tile" lever of DESIGN section 6);
    # 1 = weights (or pixels) entirely in registers
    cases = [("reg", 0, 0), ("lds", 1, 0), ("lds", 2, 0), ("lds", 3, 0), ("lds", 4, 0), ("dma", 2, 64), ("dma", 3, 64), ("dma", 4, 64)]
    for data in ("random", "zero"):
        for mode, rp4, per_kib in cases:
            cmd = [exe, "--mode", mode, "--data", data, "--seconds", str(args.seconds)]
            if rp4:
                cmd += ["--reads-per-4", str(rp4)]
            if per_kib:
                cmd += ["--mfma-per-kib", str(per_kib)]
            with GpuSampler(0, period=0.05) as smp:
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
            if r.returncode != 0:
                print("FAILED", cmd, r.stderr[-500:], file=sys.stderr)
                continue
            row = json.loads(r.stdout.strip().splitlines()[-1])
            row.update(smp.summary())
            if row["sclk_mhz_mean"]:
                # MFMA pipe utilisation at the MEASURED clock: 2500 TFLOP/s is 256 CUs x 4 SIMDs x 1017.25 flop / clk at 2400 MHz
                row["frac_of_peak_at_measured_clock"] = round(row["tflops"] / (2500.0 * row["sclk_mhz_mean"] / 2400.0), 4)
            rows.append(row)
            print(json.dumps(row), flush=True)
    bench = None
    with GpuSampler(0, period=0.05) as smp:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(args.bench_steps), "--warmup", "20", "--cpu-images", "0", "--no-other-configs"],
                           capture_output=True, text=True, timeout=600)
    if r.returncode == 0:
        line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
        b = json.loads(line)
        bench = {"images_per_s": b["value"], "conv_stack_tflops": b["roofline"]["achieved"], "frac_of_2500": b["roofline"]["frac"],
                 "issued_mfma_tflops": b["roofline"]["issued_mfma_tflops"]}
        # clock and power over the bench's OWN warm-up + timed region (its sampler brackets exactly that; the one around the subprocess also
        # sees the minute of `import torch` on a fresh box)
        bench.update({"sclk_mhz_mean": b["roofline"].get("sclk_mhz_mean"), "package_power_w_mean": b["roofline"].get("package_power_w_mean"),
                      "samples": b["roofline"].get("clock_samples"), "source": b["roofline"].get("clock_source")})
        if bench["sclk_mhz_mean"]:
            bench["frac_of_peak_at_measured_clock"] = round(bench["issued_mfma_tflops"] / (2500.0 * bench["sclk_mhz_mean"] / 2400.0), 4)
        print(json.dumps({"bench": bench}), flush=True)
    else:
        print("bench.py failed:", r.stderr[-800:], file=sys.stderr)
    lines = []
    lines.append("MFMA ceiling of this MI355X at its package power cap (tools/mfma_ceiling.py; v_mfma_f32_32x32x16_bf16 only, 256 CUs, two waves per SIMD;")
    lines.append("%.0f s per row, the first 30 %% untimed; clock / power: mean of the middle 60 %% of 50-ms sysfs samples of the card under load)" % args.seconds)
    lines.append("")
    lines.append("%-8s %-6s %-14s %-12s %10s %10s %10s %12s %14s" % ("data", "mode", "LDS reads/4", "MFMA/KiB DMA", "TFLOP/s", "of 2500", "sclk MHz", "package W", "of peak@clock"))
    for row in rows:
        lines.append("%-8s %-6s %-14s %-12s %10.1f %10.4f %10s %12s %14s" % (
            row["data"], row["mode"], row["reads_per_4_mfma"] or "-", row["mfma_per_kib"] or "-", row["tflops"], row["frac_of_2500"],
            row.get("sclk_mhz_mean"), row.get("package_power_w_mean"), row.get("frac_of_peak_at_measured_clock")))
    if bench:
        lines.append("")
        lines.append("bench.py headline on the same box right after (batch 32 at 600 x 900, bf16, %d steps): %.1f images/s, conv stack %.1f TFLOP/s algorithmic "
                     "(%.1f issued) = %.4f of 2500; sclk %s MHz, %s W; issued MFMA flops = %s of the peak at that clock"
                     % (args.bench_steps, bench["images_per_s"], bench["conv_stack_tflops"], bench["issued_mfma_tflops"], bench["frac_of_2500"],
                        bench.get("sclk_mhz_mean"), bench.get("package_power_w_mean"), bench.get("frac_of_peak_at_measured_clock")))
        rnd = {(r_["mode"], r_["reads_per_4_mfma"]): r_ for r_ in rows if r_["data"] == "random"}
        for key, what in ((("reg", 0), "the matrix pipe alone"), (("lds", 3), "MFMA + the conv kernels' LDS fragment-read mix"), (("dma", 3), "... + LDS-DMA streaming")):
            if key in rnd:
                lines.append("  conv stack (issued) / ceiling on random data, %s: %.3f" % (what, bench["issued_mfma_tflops"] / rnd[key]["tflops"]))
    lines.append("")
    lines.append("raw:")
    lines += [json.dumps(r_) for r_ in rows]
    if bench:
        lines.append(json.dumps({"bench": bench}))
    txt = "\n".join(lines) + "\n"
    print(txt)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        open(args.out, "w").write(txt)


if __name__ == "__main__":
    main()
