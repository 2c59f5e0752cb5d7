#!/bin/bash
# Board power and shader clock (rocm-smi, sampled every 0.5 s) while bench.py runs on the benchmark data and on all-zero data.
# usage: bash tools/r3_power.sh TAG  -> gpurun_out/TAG/power.txt
set -u
TAG=${1:-power}; R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
: > $OUT/power.txt
rocm-smi --showmaxpower >> $OUT/power.txt 2>&1
for v in random zero; do
  extra=""; [ $v = zero ] && extra="--zero-data"
  python $R/bench.py --steps 600 --warmup 20 --cpu-images 0 --no-other-configs --stage-events off $extra > $OUT/bench_$v.json 2> $OUT/err_$v.txt &
  BP=$!
  echo "== $v data" >> $OUT/power.txt
  while kill -0 $BP 2>/dev/null; do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -i "power\|sclk" | tr '\n' ' ' >> $OUT/power.txt; echo >> $OUT/power.txt
    sleep 0.5
  done
  wait $BP
  python3 -c "import json; d=json.load(open('$OUT/bench_$v.json')); print('bench line, $v data: %.1f images/s, %.3f ms/step' % (d['value'], d['ms_per_step']))" >> $OUT/power.txt
done
cat $OUT/power.txt
