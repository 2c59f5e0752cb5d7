# Round 6: split precision's ragged tile columns through the edge kernel (option split_edge) against the padded tile column. ONE context per
# process, alternating: contexts created later in a process that already holds several split-precision contexts (25 GB each) run up to 4 %
# slower (tools/r6_ctx_position.sh), which biases any in-process round-robin of this precision.
set -x
mkdir -p gpurun_out/r6e
export CTPN_NO_TORCH=1
for r in 1 2 3 4; do
  for v in "precision=split" "precision=split split_edge=0"; do
    timeout 200 python tools/quick_bench.py --variant "$v" --steps 30 --rounds 1 2>&1 | grep -E "^round"
  done
done > gpurun_out/r6e/ab_split_edge.txt 2>&1
cat gpurun_out/r6e/ab_split_edge.txt
