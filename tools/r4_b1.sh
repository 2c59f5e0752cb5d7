#!/bin/bash
# batch-1 step timeline (rocprofv3 --kernel-trace): bash tools/r4_b1.sh TAG
TAG=${1:-b1}; R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/raw -o trace -- python $R/bench.py --batch 1 --steps 30 --warmup 5 --cpu-images 0 --no-other-configs --stage-events off > $OUT/bench.json 2> $OUT/trace.err
cd $R
python tools/timeline.py $OUT/raw/trace_results.db 12 > $OUT/timeline.txt 2>&1
rm -rf $OUT/raw
cat $OUT/timeline.txt | cut -c1-150
python -c "import json; d=json.loads([l for l in open('$OUT/bench.json') if l.startswith('{')][0]); print('batch1 under trace', d['value'], d['ms_per_step'])"
