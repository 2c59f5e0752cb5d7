#!/bin/bash
# Round 5: determinism stress of the final tree (fresh ctxs, repeated calls: bit-identical heads and rois) -- batch 8 / 3 / 2 (1280 x 1920) as in
# round 4, plus batches of 1, 2 and 4 at 600 x 900 (the segmented sort + rank merge and the one-column-per-wave NMS, with their ticket and
# self-clearing survivor mask, run there) -- and the option matrix.
OUT=gpurun_out/${1:-stress}; mkdir -p $OUT
(timeout 300 python tests/gpu_stress.py bf16 12 8 600 900; timeout 200 python tests/gpu_stress.py fp16 8 3 101 203; timeout 200 python tests/gpu_stress.py bf16 8 2 1280 1920;
 timeout 100 python tests/gpu_stress.py bf16 30 1 600 900; timeout 100 python tests/gpu_stress.py split 6 2 600 900; timeout 100 python tests/gpu_stress.py bf16 10 4 600 900) 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tee $OUT/stress.txt
bash tools/switch_matrix.sh 2>&1 | tee $OUT/switch_matrix.txt
