#!/bin/bash
# Collect the judged artefacts of one round on the GPU box in ONE gpurun call (run from the repo root):
#   gpurun --timeout 1500 -- 'bash tools/profile_round.sh r01'
# Writes gpurun_out/<tag>/{bench_n1,bench_under_trace,bench_fp32_b8,bench_hires_b8_O,bench_host_images,bench_b1,bench_lstm_split}.json,
# kernel_stats.csv, layers.csv, pmc.json, pytest_gpu.txt; copy what is judged into profiles/<tag>_*.
# rocprofv3: kernel trace and every PMC group in its own pass (never combined with sys/hip/hsa tracing).
set -u
TAG=${1:-r02}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
STEPS=4
timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -3 > $OUT/pytest_gpu.txt
python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
python bench.py --precision fp32 --batch 8 --cpu-images 0 > $OUT/bench_fp32_b8.json 2>> $OUT/bench_n1.err
python bench.py --batch 8 --height 1280 --width 1920 --mode O --cpu-images 0 > $OUT/bench_hires_b8_O.json 2>> $OUT/bench_n1.err
python bench.py --host-images --cpu-images 0 > $OUT/bench_host_images.json 2>> $OUT/bench_n1.err
python bench.py --batch 1 --steps 200 --warmup 20 --cpu-images 0 > $OUT/bench_b1.json 2>> $OUT/bench_n1.err
python bench.py --lstm-split --cpu-images 0 > $OUT/bench_lstm_split.json 2>> $OUT/bench_n1.err
timeout 400 python tests/accuracy_report.py --images 32 --out $OUT/accuracy.json > /dev/null 2>> $OUT/bench_n1.err
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/raw -o trace -- python $R/bench.py --steps 8 --warmup 2 --cpu-images 0 > $OUT/bench_under_trace.json 2> $OUT/trace.err
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
  name=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace -d $OUT/raw -o pmc_$name -- python $R/bench.py --steps $STEPS --warmup 1 --cpu-images 0 --stage-events off > /dev/null 2> $OUT/pmc_$name.err
done
cd $R
python tools/rocprof_summary.py $OUT/raw/trace_results.db $OUT/kernel_stats.csv
python tools/rocprof_layers.py $OUT/raw/trace_results.db $OUT/layers.csv > $OUT/layers.txt
python tools/timeline.py $OUT/raw/trace_results.db 3 > $OUT/timeline.txt 2>&1
python tools/pmc_summary.py $OUT/raw/pmc_FETCH_SIZE_results.db $OUT/raw/pmc_WRITE_SIZE_results.db $OUT/raw/pmc_SQ_VALU_MFMA_BUSY_CYCLES_results.db $((STEPS + 1)) $OUT/pmc.json
rm -rf $OUT/raw        # databases are large; the summaries are what travels back
ls -la $OUT
