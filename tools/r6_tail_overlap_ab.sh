# Round 6: option tail_overlap (BiLSTM + heads of batch k on the proposal stream, beside conv1_1 of batch k + 1) in split precision, where conv1_1 is a
# stand-alone 1.08 ms HBM-write-bound kernel: same-box A/B (each variant twice, interleaved, to see the order effect), then the two-batches-in-flight check
set -x
mkdir -p gpurun_out/r6e
export CTPN_NO_TORCH=1
timeout 600 python tools/quick_bench.py --variant "precision=split tail_overlap=1" --variant "precision=split" --variant "precision=split tail_overlap=1 nms_prefix=1" --variant "precision=split nms_prefix=1" --steps 40 --rounds 3 --stages 2>&1 | grep -E "^round|\"(conv_first|conv_gemm|lstm|nms|bilstm|heads|fc)" > gpurun_out/r6e/ab_tail_overlap.txt; cat gpurun_out/r6e/ab_tail_overlap.txt
timeout 300 python tools/r6_pipeline_race.py --reps 16 --variants "split:tail_overlap=1" "bf16:tail_overlap=1" 2>&1 | cut -c1-200 | tee -a gpurun_out/r6e/ab_tail_overlap.txt
