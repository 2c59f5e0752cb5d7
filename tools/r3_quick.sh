#!/bin/bash
# quick GPU iteration: a pytest subset (arg 2, -k expression; empty = skip), the default bench without the CPU legs, per-layer trace
# usage: bash tools/r3_quick.sh TAG "pytest -k expr" [extra pytest args]
set -u
TAG=$1; KEXPR=${2:-}
R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
if [ -n "$KEXPR" ]; then (time timeout 900 python -m pytest tests -m gpu -q -k "$KEXPR" 2>&1 | tail -40) > $OUT/pytest.txt 2>&1; tail -12 $OUT/pytest.txt; fi
python bench.py --cpu-images 0 --no-other-configs > $OUT/bench.json 2> $OUT/bench.err
python -c "import json;d=json.load(open('$OUT/bench.json'));print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['stages_ms_per_step'])" || tail -5 $OUT/bench.err
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/raw -o trace -- python $R/bench.py --steps 8 --warmup 2 --cpu-images 0 --no-other-configs > $OUT/bench_under_trace.json 2> $OUT/trace.err
cd $R
python tools/rocprof_layers.py $OUT/raw/trace_results.db $OUT/layers.csv > $OUT/layers.txt 2>&1
rm -rf $OUT/raw
cut -d, -f1,2,4 $OUT/layers.csv | sed 's/_ZN4ctpn//' | cut -c1-90
