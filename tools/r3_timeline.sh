#!/bin/bash
# one bench step as a timeline under the given env: bash tools/r3_timeline.sh TAG ENV=VAL...
TAG=$1; shift; R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
cd /tmp
rocprofv3 --kernel-trace -d $OUT/raw -o trace -- python $R/bench.py --steps 6 --warmup 2 --cpu-images 0 --no-other-configs > $OUT/bench.json 2> $OUT/err.txt
cd $R
python tools/timeline.py $OUT/raw/trace_results.db 3 > $OUT/timeline.txt 2>&1
rm -rf $OUT/raw
cat $OUT/timeline.txt
