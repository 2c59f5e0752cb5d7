#!/bin/bash
# Round 4: the two judged bench lines of the final tree (the default command, and fp16w), written to gpurun_out/<tag>/
set -u
R=$PWD; OUT=$R/gpurun_out/${1:-r04c}; mkdir -p $OUT
python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
python bench.py --precision fp16w --no-other-configs > $OUT/bench_fp16w.json 2>> $OUT/bench_n1.err
python - $OUT/bench_n1.json $OUT/bench_fp16w.json <<'PY'
import json, sys
for f in sys.argv[1:]:
    d = json.loads([l for l in open(f) if l.startswith("{")][0])
    r = d["roofline"]
    print(f.split("/")[-1], d["value"], d["ms_per_step"], r["frac"], r["flops_per_image"], r["conv1_1_in_family"], r["traffic"])
    for k, v in d.get("other_configs", {}).items():
        if isinstance(v, dict): print("   ", k, v.get("images_per_s"), v.get("ms_per_step"), v.get("sclk_mhz_mean"), v.get("package_power_w_mean"))
PY
