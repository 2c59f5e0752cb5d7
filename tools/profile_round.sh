#!/bin/bash
# Collect the judged artefacts of one round on the GPU box in ONE gpurun call (run from the repo root):
#   gpurun --timeout 2400 -- 'bash tools/profile_round.sh r06'
# Writes gpurun_out/<tag>/{pytest_gpu.txt, bench_n1.json (the default line: headline + accuracy + cpu_baseline + other_configs),
# decode_throughput_kinds.json (files in / lines out per file kind), kernel_stats.csv, layers.csv / layers.txt, timeline.txt (bf16), timeline_sync.txt + latency.json (one
# synchronous single-image call, tools/r5_latency.sh), pmc.json};
# copy what is judged into profiles/<tag>_*.
# rocprofv3: kernel trace and every PMC group in its own pass (never combined with sys/hip/hsa tracing).
set -u
TAG=${1:-r06}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
STEPS=4
(time timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -8) > $OUT/pytest_gpu.txt 2>&1
python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
CTPN_NO_TORCH=1 timeout 120 python tools/file_kinds_throughput.py --images 768 --distinct 64 --out $OUT/decode_throughput_kinds.json > /dev/null 2> $OUT/kinds.err
cd /tmp
for p in bf16; do
  rocprofv3 --kernel-trace --stats -d $OUT/raw_$p -o trace -- python $R/bench.py --precision $p --steps 8 --warmup 2 --cpu-images 0 --no-other-configs > $OUT/bench_under_trace_$p.json 2> $OUT/trace_$p.err
done
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
  name=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace -d $OUT/raw -o pmc_$name -- python $R/bench.py --steps $STEPS --warmup 1 --cpu-images 0 --no-other-configs --stage-events off > /dev/null 2> $OUT/pmc_$name.err
done
cd $R
python tools/rocprof_summary.py $OUT/raw_bf16/trace_results.db $OUT/kernel_stats.csv
python tools/rocprof_layers.py $OUT/raw_bf16/trace_results.db $OUT/layers.csv > $OUT/layers.txt
python tools/timeline.py $OUT/raw_bf16/trace_results.db 3 > $OUT/timeline.txt 2>&1
python tools/pmc_summary.py $OUT/raw/pmc_FETCH_SIZE_results.db $OUT/raw/pmc_WRITE_SIZE_results.db $OUT/raw/pmc_SQ_VALU_MFMA_BUSY_CYCLES_results.db $((STEPS + 1)) $OUT/pmc.json
rm -rf $OUT/raw $OUT/raw_bf16        # databases are large; the summaries are what travels back
bash tools/r5_latency.sh $TAG > /dev/null 2>&1
# round 6: the drop-in's default precision (split) profiled like the headline -- per-layer times, one synchronous single-image call -- the MFMA
# ceiling of this box next to one bench run (tools/mfma_ceiling.py), and the two-batches-in-flight check of every precision
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/raw_split -o trace -- python $R/bench.py --precision split --steps 4 --warmup 1 --cpu-images 0 --no-other-configs > $OUT/bench_under_trace_split.json 2> $OUT/trace_split.err
cd $R
python tools/rocprof_layers.py $OUT/raw_split/trace_results.db $OUT/layers_split.csv > $OUT/layers_split.txt
python tools/timeline.py $OUT/raw_split/trace_results.db 3 > $OUT/timeline_split.txt 2>&1
rm -rf $OUT/raw_split
bash tools/r5_latency.sh $TAG/split1 --precision split > /dev/null 2>&1
python tools/mfma_ceiling.py --seconds 4 --out $OUT/mfma_ceiling.txt > $OUT/mfma_ceiling.log 2>&1
CTPN_NO_TORCH=1 python tools/r6_pipeline_race.py --reps 30 --variants split: bf16: fp16: fp32: 2>&1 | grep -v "^RCCL\|amdgpu.ids" | cut -c1-400 > $OUT/pipeline_race.txt
# the stress that exposed the barrier race (two batches in flight, the first one's NMS delayed into conv3_x .. conv5_x of the second): 0 mismatches expected
ROUNDS=1 bash tools/r6_barrier_war.sh > /dev/null 2>&1
cp gpurun_out/r6n/war_fix_race.txt $OUT/barrier_war_stress.txt
ls -la $OUT
