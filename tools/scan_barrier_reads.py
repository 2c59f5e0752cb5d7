#!/usr/bin/env python
"""Static check of the device code (round 6): is any LDS read in flight across an s_barrier?

The conv kernels recycle LDS buffers with LDS-DMA issued right behind a raw s_barrier; a ds_read issued in front of the barrier whose
s_waitcnt lgkmcnt hipcc has sunk BELOW it is then ordered against that DMA by latency only (profiles/r06_barrier_war.txt: it lost under
memory-system load from another stream). This script compiles the named .hip files to gfx950 assembly and reports, per kernel, the barriers
that have ds_read instructions between the last `s_waitcnt lgkmcnt(0)` (or the previous barrier / a label) and the barrier.

    python tools/scan_barrier_reads.py [files ...]        (default: every csrc/*.hip that issues LDS-DMA)
Exit code 1 if a kernel on the list MUST_BE_CLEAN has such a barrier.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "text-detection-ctpn_amd", "csrc")
FLAGS = {"conv3x3": ["-fno-honor-nans", "-mno-amdgpu-ieee"]}
MUST_BE_CLEAN = ("conv3x3_p_kernel", "conv3x3_kernel")


def scan(asm):
    kern, rows = None, {}
    for line in asm.split("\n"):
        m = re.match(r"^(_Z\S+):", line)
        if m:
            kern = m.group(1)
            rows[kern] = []
            continue
        t = line.strip()
        if kern is None or not t or t.startswith(";") or (t.startswith(".") and not t.startswith(".LBB")):
            continue
        rows[kern].append(t)
    out = {}
    for k, v in rows.items():
        nb = bad = 0
        for i, t in enumerate(v):
            if not t.startswith("s_barrier"):
                continue
            nb += 1
            pend = 0
            for j in range(i - 1, -1, -1):
                tt = v[j]
                if tt.startswith(".LBB") or tt.startswith("s_barrier") or (tt.startswith("s_waitcnt") and "lgkmcnt(0)" in tt):
                    break
                if tt.startswith("ds_read"):
                    pend += 1
            bad += pend > 0
        if nb:
            out[k] = (nb, bad)
    return out


def main():
    files = sys.argv[1:] or [f for f in sorted(os.listdir(CSRC)) if f.endswith(".hip") and (f.startswith("conv3x3_") or "global_load_lds" in open(os.path.join(CSRC, f)).read())]
    rc = 0
    for f in files:
        src = os.path.join(CSRC, os.path.basename(f))
        extra = [x for key, fl in FLAGS.items() if os.path.basename(f).startswith(key) for x in fl]
        with tempfile.TemporaryDirectory() as td:
            o = os.path.join(td, "a.s")
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-inline-asm", "-S", "--cuda-device-only", "-o", o, src] + extra,
                           check=True, stderr=subprocess.DEVNULL)
            res = scan(open(o).read())
        for k, (nb, bad) in res.items():
            name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
            name = re.sub(r"\(ctpn::\w+\)$", "", name)
            flag = ""
            if bad and any(m in name for m in MUST_BE_CLEAN):
                flag = "   <-- MUST BE CLEAN"
                rc = 1
            print("%-22s %-150s barriers %3d   with LDS reads in flight %3d%s" % (os.path.basename(f), name[:150], nb, bad, flag))
    return rc


if __name__ == "__main__":
    sys.exit(main())
