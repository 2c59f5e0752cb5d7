#!/bin/bash
# One SYNCHRONOUS single-image call (ctpn_detect, nothing in flight) as a kernel timeline (rocprofv3 --kernel-trace): bash tools/r5_latency.sh TAG [bench options]
TAG=${1:-lat}; shift; R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/raw -o trace -- python $R/bench.py --latency-only "$@" > $OUT/latency_under_trace.json 2> $OUT/trace.err
cd $R
python tools/timeline.py $OUT/raw/trace_results.db 60 > $OUT/timeline_sync.txt 2>&1
rm -rf $OUT/raw
cut -c1-150 $OUT/timeline_sync.txt
python bench.py --latency-only "$@" > $OUT/latency.json 2>> $OUT/trace.err
cat $OUT/latency.json
