#!/bin/bash
# round 3, first GPU call: the whole GPU suite, the default bench line (with accuracy + other_configs), per-layer trace
set -u
R=$PWD; OUT=$R/gpurun_out/r03a; mkdir -p $OUT
export TMPDIR=/tmp
(time timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -40) > $OUT/pytest_gpu.txt 2>&1
tail -5 $OUT/pytest_gpu.txt
(time python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err); tail -3 $OUT/bench_n1.err
python - <<PY
import json
d=json.load(open("$OUT/bench_n1.json"))
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"])
print("accuracy", d.get("accuracy"))
print("other", json.dumps(d.get("other_configs"), indent=0)[:3000])
PY
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/raw -o trace -- python $R/bench.py --steps 8 --warmup 2 --cpu-images 0 --no-other-configs > $OUT/bench_under_trace.json 2> $OUT/trace.err
cd $R
python tools/rocprof_layers.py $OUT/raw/trace_results.db $OUT/layers.csv > $OUT/layers.txt 2>&1
python tools/rocprof_summary.py $OUT/raw/trace_results.db $OUT/kernel_stats.csv > /dev/null 2>&1
rm -rf $OUT/raw
cut -d, -f1,2,4 $OUT/layers.csv | sed 's/_ZN4ctpn//' | cut -c1-110
