# Round 6: split precision's ragged tile columns through the edge kernel (option split_edge) against the padded tile column: same-box A/B
set -x
mkdir -p gpurun_out/r6e
export CTPN_NO_TORCH=1
timeout 600 python tools/quick_bench.py --variant "precision=split" --variant "precision=split split_edge=0" --steps 30 --rounds 3 --stages 2>&1 | grep -E "^round|conv_gemm|conv_first" | head -20 > gpurun_out/r6e/ab_split_edge.txt; cat gpurun_out/r6e/ab_split_edge.txt
if [ "${TESTS:-1}" = 1 ]; then python -m pytest tests -m gpu -q -x -k "split or precision or round6 or lone or batch" > gpurun_out/r6e/pytest_split.txt 2>&1; tail -4 gpurun_out/r6e/pytest_split.txt; fi
