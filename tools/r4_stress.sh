#!/bin/bash
# Round 4: determinism stress of the fused tree (fresh ctxs, repeated calls: bit-identical heads and rois) + the option matrix.
OUT=gpurun_out/${1:-stress}; mkdir -p $OUT
(timeout 300 python tests/gpu_stress.py bf16 12 8 600 900; timeout 200 python tests/gpu_stress.py fp16 8 3 101 203; timeout 200 python tests/gpu_stress.py bf16 8 2 1280 1920; timeout 100 python tests/gpu_stress.py bf16 20 1 600 900) 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tee $OUT/stress.txt
bash tools/switch_matrix.sh 2>&1 | tee $OUT/switch_matrix.txt
