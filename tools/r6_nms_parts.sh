# Which part of the one-workgroup proposal NMS disturbs the next batch's split-precision layers? (diagnostic option debug_nms; round 6)
set -x
mkdir -p gpurun_out/r6n
export CTPN_NO_TORCH=1
B="split:tail_confine=0,nms_prefix=0"
timeout 1100 python tools/r6_pipeline_race.py --reps 24 --batch 32 --heads --variants "$B" "$B,debug_nms=2" "$B,debug_nms=4" "$B,debug_nms=6" "$B,debug_nms=1" "$B,debug_nms=8" \
   "$B,debug_hog=1200,debug_nms=1" "$B,debug_hog=1200,debug_nms=8" > gpurun_out/r6n/parts.txt 2>&1
grep -E "heads" gpurun_out/r6n/parts.txt | cut -c1-200
