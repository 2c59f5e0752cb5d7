#!/bin/bash
# Round 6, VERDICT r5 "next" 2: the split mode's first layers. Same-box A/B of conv1_2 through the persistent kernel's 64-channel form
# (option conv_p64 = 1, default) against the non-persistent kernel (conv_p64 = 0), split precision at batch 32 and fp32 at batch 8, then the
# per-layer profile of the split step. Run from the repo root:   gpurun --timeout 900 -- 'bash tools/r6_split_ab.sh'
R=$PWD; OUT=$R/gpurun_out/r6split; mkdir -p $OUT; export TMPDIR=/tmp
CTPN_NO_TORCH=1 python tools/quick_bench.py --variant "precision=split" --variant "precision=split conv_p64=0" --steps 20 --rounds 3 --stages --out $OUT/ab_split.json > $OUT/ab_split.txt 2>&1
CTPN_NO_TORCH=1 python tools/quick_bench.py --batch 8 --variant "precision=fp32" --variant "precision=fp32 conv_p64=0" --steps 6 --rounds 3 --out $OUT/ab_fp32.json > $OUT/ab_fp32.txt 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/raw -o trace -- python $R/bench.py --precision split --steps 4 --warmup 1 --cpu-images 0 --no-other-configs --stage-events off > $OUT/bench.json 2> $OUT/err.txt
cd $R; python tools/rocprof_layers.py $OUT/raw/trace_results.db $OUT/layers.csv > $OUT/layers.txt; rm -rf $OUT/raw
grep "^round\|images_per_s" $OUT/ab_split.txt | head -20; grep "^round" $OUT/ab_fp32.txt; cut -c1-160 $OUT/layers.txt | head -60
