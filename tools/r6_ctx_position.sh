# Round 6: identical contexts in one process do not run at identical speed (tools/quick_bench.py round-robin: +-1.5 %). Is it the placement of their
# buffers? Four default-equal variants per precision, round-robin, three rounds.
set -x
mkdir -p gpurun_out/r6p
export CTPN_NO_TORCH=1
timeout 600 python tools/quick_bench.py --variant "precision=split" --variant "precision=split nms_prefix=1" --variant "precision=split conv_p64=1" --variant "precision=split split_edge=1" --steps 30 --rounds 3 2>&1 | grep -E "^round" > gpurun_out/r6p/ctx_position.txt
timeout 600 python tools/quick_bench.py --variant "precision=bf16" --variant "precision=bf16 nms_prefix=1" --variant "precision=bf16 conv1_fuse=1" --variant "precision=bf16 lstm_split=1" --steps 60 --rounds 3 2>&1 | grep -E "^round" >> gpurun_out/r6p/ctx_position.txt
cat gpurun_out/r6p/ctx_position.txt
