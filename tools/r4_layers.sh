#!/bin/bash
# Per-launch kernel durations of one bench step for the given precisions (rocprofv3 --kernel-trace): bash tools/r4_layers.sh TAG fp16 fp16w
TAG=${1:-lay}; shift; R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for p in "$@"; do
  cd /tmp
  rocprofv3 --kernel-trace --stats -d $OUT/raw_$p -o trace -- python $R/bench.py --precision $p --steps 6 --warmup 2 --cpu-images 0 --no-other-configs > $OUT/bench_$p.json 2> $OUT/trace_$p.err
  cd $R
  python tools/rocprof_summary.py $OUT/raw_$p/trace_results.db $OUT/kernel_stats_$p.csv
  python tools/rocprof_layers.py $OUT/raw_$p/trace_results.db $OUT/layers_$p.csv > $OUT/layers_$p.txt 2>&1
  python tools/timeline.py $OUT/raw_$p/trace_results.db 3 > $OUT/timeline_$p.txt 2>&1
  rm -rf $OUT/raw_$p
  echo "== $p"; cat $OUT/layers_$p.txt | head -40
done
