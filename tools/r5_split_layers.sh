R=$PWD; OUT=$R/gpurun_out/r5split; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/raw -o trace -- python $R/bench.py --precision split --steps 4 --warmup 1 --cpu-images 0 --no-other-configs --stage-events off > $OUT/bench.json 2> $OUT/err.txt
cd $R; python tools/rocprof_layers.py $OUT/raw/trace_results.db $OUT/layers.csv > $OUT/layers.txt; rm -rf $OUT/raw; cut -c1-150 $OUT/layers.txt
