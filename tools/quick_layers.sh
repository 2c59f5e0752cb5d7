#!/bin/bash
# One gpurun call: a subset of the GPU tests, the default bench (+ A/B variants given as "NAME:ENV=VAL,..." arguments) and a
# kernel trace of the default configuration summarised per layer.   usage: bash tools/quick_layers.sh TAG "pytest -k expr" [variants...]
set -u
TAG=$1; KEXPR=$2; shift 2
R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
if [ -n "$KEXPR" ]; then (time python -m pytest tests -m gpu -x -q -k "$KEXPR" 2>&1 | tail -25) > $OUT/pytest.txt 2>&1; tail -4 $OUT/pytest.txt; fi
python bench.py --cpu-images 0 > $OUT/bench_default.json 2> $OUT/bench_default.err; python -c "import json;d=json.load(open('$OUT/bench_default.json'));print('default', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['stages_ms_per_step'])"
for v in "$@"; do
  name=${v%%:*}; envs=${v#*:}
  ( IFS=,; for kv in $envs; do export "$kv"; done; python $R/bench.py --cpu-images 0 > $OUT/bench_$name.json 2> $OUT/bench_$name.err )
  python -c "import json;d=json.load(open('$OUT/bench_$name.json'));print('$name', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['stages_ms_per_step'])" || tail -3 $OUT/bench_$name.err
done
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/raw -o trace -- python $R/bench.py --steps 8 --warmup 2 --cpu-images 0 > $OUT/bench_under_trace.json 2> $OUT/trace.err
cd $R
python tools/rocprof_layers.py $OUT/raw/trace_results.db $OUT/layers.csv > $OUT/layers.txt 2>&1
python tools/rocprof_summary.py $OUT/raw/trace_results.db $OUT/kernel_stats.csv > /dev/null 2>&1
python tools/timeline.py $OUT/raw/trace_results.db > $OUT/timeline.txt 2>&1
rm -rf $OUT/raw
cut -d, -f1,2,4 $OUT/layers.csv | sed 's/_ZN4ctpn//' | cut -c1-110
