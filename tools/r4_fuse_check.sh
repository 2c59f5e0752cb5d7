#!/bin/bash
# Round 4: the fused conv1_1 / conv1_2 launch on the GPU: the byte-equality tests, then a same-box A/B of the fused and the stored form
# (options conv1_fuse = 1 | 0). Everything under its own timeout (a persistent kernel that deadlocks must not take the box with it).
# Usage: tools/r4_fuse_check.sh [outdir]. The A/B against rounds 2-3's conv_first_q_kernel (removed since) is profiles/r04_ab_conv1_fuse.txt.
OUT=${1:-gpurun_out/fuse}
mkdir -p "$OUT"
timeout 420 python -m pytest tests/test_gpu_fuse.py -x -q > "$OUT/pytest_fuse.txt" 2>&1
echo "pytest exit $?" >> "$OUT/pytest_fuse.txt"
tail -5 "$OUT/pytest_fuse.txt"
B="python bench.py --steps 30 --warmup 10 --cpu-images 0 --no-other-configs"
CTPN_CONV1_FUSE=1 timeout 200 $B > "$OUT/bench_fused.json" 2> "$OUT/bench_fused.err"
CTPN_CONV1_FUSE=0 timeout 200 $B > "$OUT/bench_stored_q.json" 2> "$OUT/bench_stored_q.err"
CTPN_CONV1_FUSE=1 timeout 200 $B > "$OUT/bench_fused_b.json" 2> "$OUT/bench_fused_b.err"
for f in fused stored_q fused_b; do
  python - "$OUT/bench_$f.json" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[1], d["value"], d["ms_per_step"], d.get("roofline", {}).get("frac"), d.get("stages_ms_per_step"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
