# Round 6: the one-workgroup proposal NMS of batch A delayed (debug_hog) into later layers of batch B: which layer's output changes first, and how?
set -x
mkdir -p gpurun_out/r6n
export CTPN_NO_TORCH=1
L="conv3_2 conv3_3 pool3 conv4_1 conv4_2 conv4_3 pool4 conv5_1"
for k in 1 2 3 4; do
timeout 500 python tools/r6_pipeline_race.py --reps 12 --batch 32 --diagnose --dump gpurun_out/r6n/patch$k.npz --layers $L --variants "bf16:nms_prefix=0,debug_hog=7000" 2>&1 | grep diagnose | cut -c1-1500
done > gpurun_out/r6n/bf16_diag2.txt 2>&1
cat gpurun_out/r6n/bf16_diag2.txt; ls -la gpurun_out/r6n/
