set -x
mkdir -p gpurun_out/r6l
export CTPN_NO_TORCH=1
timeout 1500 python tools/r6_pipeline_race.py --reps 300 --batch 32 --variants "bf16:nms_prefix=0,debug_hog=5000" "bf16:nms_prefix=0,debug_hog=7000" "split:tail_confine=0,nms_prefix=0" "split:" "bf16:" 2>&1 | cut -c1-200 > gpurun_out/r6l/long_stress.txt
cat gpurun_out/r6l/long_stress.txt
