# Round 6: do identical contexts of one process run at identical speed? Four default-equal variants per precision, round-robin, three rounds.
# (They did not while split precision forked its edge kernels onto the process-wide helper stream: up to 4.5 % apart. Final tree: within 0.3 %.)
set -x
mkdir -p gpurun_out/r6p
export CTPN_NO_TORCH=1
timeout 600 python tools/quick_bench.py --variant "precision=split" --variant "precision=split nms_prefix=1" --variant "precision=split conv_p64=1" --variant "precision=split split_edge=1" --steps 30 --rounds 3 2>&1 | grep -E "^round" > gpurun_out/r6p/ctx_position.txt
timeout 600 python tools/quick_bench.py --variant "precision=bf16" --variant "precision=bf16 nms_prefix=1" --variant "precision=bf16 conv1_fuse=1" --variant "precision=bf16 lstm_split=1" --steps 60 --rounds 3 2>&1 | grep -E "^round" >> gpurun_out/r6p/ctx_position.txt
cat gpurun_out/r6p/ctx_position.txt
