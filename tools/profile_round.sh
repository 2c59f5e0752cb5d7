#!/bin/bash
# Collect the judged artefacts of one round on the GPU box in ONE gpurun call (run from the repo root):
#   gpurun --timeout 1500 -- 'bash tools/profile_round.sh r03'
# Writes gpurun_out/<tag>/{pytest_gpu.txt, bench_n1.json (the default line: headline + accuracy + cpu_baseline + other_configs),
# bench_lstm_split.json, accuracy.json (32 images), kernel_stats.csv, layers.csv, timeline.txt, pmc.json, pmc_wr_global_ranges.txt};
# copy what is judged into profiles/<tag>_*.
# rocprofv3: kernel trace and every PMC group in its own pass (never combined with sys/hip/hsa tracing).
set -u
TAG=${1:-r03}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
STEPS=4
(time timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -8) > $OUT/pytest_gpu.txt 2>&1
python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
python bench.py --lstm-split --cpu-images 0 --no-other-configs > $OUT/bench_lstm_split.json 2>> $OUT/bench_n1.err
timeout 500 python tests/accuracy_report.py --images 32 --out $OUT/accuracy.json > /dev/null 2>> $OUT/bench_n1.err
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/raw -o trace -- python $R/bench.py --steps 8 --warmup 2 --cpu-images 0 --no-other-configs > $OUT/bench_under_trace.json 2> $OUT/trace.err
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
  name=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace -d $OUT/raw -o pmc_$name -- python $R/bench.py --steps $STEPS --warmup 1 --cpu-images 0 --no-other-configs --stage-events off > /dev/null 2> $OUT/pmc_$name.err
done
# A/B of the XCD-local tile ranges of conv3x3_wr_kernel: HBM reads with one grid-wide range (round 2's mapping)
CTPN_C3_WR_XCD=0 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/raw -o pmc_FETCH_xcd0 -- python $R/bench.py --steps $STEPS --warmup 1 --cpu-images 0 --no-other-configs --stage-events off > /dev/null 2> $OUT/pmc_FETCH_xcd0.err
cd $R
python tools/rocprof_summary.py $OUT/raw/trace_results.db $OUT/kernel_stats.csv
python tools/rocprof_layers.py $OUT/raw/trace_results.db $OUT/layers.csv > $OUT/layers.txt
python tools/timeline.py $OUT/raw/trace_results.db 3 > $OUT/timeline.txt 2>&1
python tools/pmc_summary.py $OUT/raw/pmc_FETCH_SIZE_results.db $OUT/raw/pmc_WRITE_SIZE_results.db $OUT/raw/pmc_SQ_VALU_MFMA_BUSY_CYCLES_results.db $((STEPS + 1)) $OUT/pmc.json
python - <<PY > $OUT/pmc_wr_global_ranges.txt
import sqlite3
db = sqlite3.connect("$OUT/raw/pmc_FETCH_xcd0_results.db")
print("HBM read bytes per launch with CTPN_C3_WR_XCD=0 (one tile range for the whole grid, round 2's mapping); FETCH_SIZE x 2 KiB")
for name, val, cnt in db.execute("select kernel_name, sum(value), count(*) from counters_collection where counter_name = 'FETCH_SIZE' group by kernel_name"):
    if "conv3x3_wr" in name:
        print("%-70s %.1f MB (%d launches)" % (name[:70], val * 2048 / cnt / 1e6, cnt))
PY
rm -rf $OUT/raw        # databases are large; the summaries are what travels back
ls -la $OUT
