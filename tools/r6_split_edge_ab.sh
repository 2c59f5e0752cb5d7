# Round 6: split precision's ragged tile columns through the edge kernel (option split_edge) against the padded tile column. ONE context per
# process, alternating: with the edge kernels forked onto the process-wide helper stream (the form this tool measured first) default-equal
# split contexts of one process ran up to 4.5 % apart (tools/r6_ctx_position.sh), which biases an in-process round-robin.
set -x
mkdir -p gpurun_out/r6e
export CTPN_NO_TORCH=1
for r in 1 2 3 4; do
  for v in "precision=split" "precision=split split_edge=0"; do
    timeout 200 python tools/quick_bench.py --variant "$v" --steps 30 --rounds 1 2>&1 | grep -E "^round"
  done
done > gpurun_out/r6e/ab_split_edge.txt 2>&1
cat gpurun_out/r6e/ab_split_edge.txt
