#!/bin/bash
# Round 6: determinism stress of the final tree (fresh ctxs, repeated calls: bit-identical heads and rois) -- round 5's set plus one image per
# call in split precision (few-rows recurrence, 64-pixel flat items, half-tile tails for 8 x 32 patches, persistent split conv1_2) -- the
# two-batches-in-flight check of every precision, and the option matrix.   gpurun --timeout 2400 -- 'bash tools/r6_stress.sh r6stress'
OUT=gpurun_out/${1:-r6stress}; mkdir -p $OUT
(timeout 300 python tests/gpu_stress.py bf16 12 8 600 900; timeout 200 python tests/gpu_stress.py fp16 8 3 101 203; timeout 200 python tests/gpu_stress.py bf16 8 2 1280 1920;
 timeout 100 python tests/gpu_stress.py bf16 30 1 600 900; timeout 100 python tests/gpu_stress.py split 6 2 600 900; timeout 100 python tests/gpu_stress.py bf16 10 4 600 900;
 timeout 100 python tests/gpu_stress.py split 12 1 600 900; timeout 100 python tests/gpu_stress.py fp16 12 1 600 900; timeout 200 python tests/gpu_stress.py split 4 1 1280 1920) 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tee $OUT/stress.txt
CTPN_NO_TORCH=1 python tools/r6_pipeline_race.py --reps 30 --variants split: bf16: fp16: fp32: 2>&1 | grep -v "^RCCL\|amdgpu.ids" | cut -c1-300 | tee $OUT/pipeline_race.txt
bash tools/switch_matrix.sh 2>&1 | tee $OUT/switch_matrix.txt
