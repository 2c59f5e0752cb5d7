#!/bin/bash
# Same-box A/B of per-ctx options on the default benchmark line: bash tools/r4_ab.sh TAG "conv1_overlap=1" "tail_overlap=1" ...
# (first run = defaults; every run: bench.py --steps 60, no CPU baseline / other configs) -> gpurun_out/TAG/ab.txt
TAG=${1:-ab}; shift; R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; : > $OUT/ab.txt
for rep in 1 2; do
for o in "" "$@"; do
  args=""; for kv in $o; do args="$args --option $kv"; done
  python $R/bench.py --steps 60 --warmup 5 --cpu-images 0 --no-other-configs $args > $OUT/b.json 2> $OUT/b.err
  python3 - "$o" $OUT/b.json >> $OUT/ab.txt <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][0])
st = d["stages_ms_per_step"]
print("%-28s %8.1f images/s  %7.3f ms/step  conv stack %6.1f TF (%.4f)  conv_first %.3f  gemm %.3f  bilstm %.3f" % (
    sys.argv[1] or "(defaults)", d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["frac"], st.get("conv_first", 0), st.get("gemm", 0), st.get("bilstm", 0)))
PY
done
done
cat $OUT/ab.txt
