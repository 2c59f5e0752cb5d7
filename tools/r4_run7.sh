set -u
R=$PWD; OUT=$R/gpurun_out/fuse7; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_round4.py tests/test_gpu_fuse.py "tests/test_gpu_parity.py::test_batch_equals_singles_and_is_idempotent" tests/test_gpu_round3.py -q -x 2>&1 | tail -4
python bench.py --batch 1 --steps 300 --warmup 20 --cpu-images 0 --no-other-configs --stage-events off > $OUT/b1.json 2>/dev/null
python -c "import json; d=json.loads([l for l in open('$OUT/b1.json') if l.startswith('{')][0]); print('batch1', d['value'], d['ms_per_step'])"
for i in 1 2; do python bench.py --steps 40 --warmup 10 --cpu-images 0 --no-other-configs > $OUT/b32_$i.json 2>/dev/null
python -c "import json; d=json.loads([l for l in open('$OUT/b32_$i.json') if l.startswith('{')][0]); print('batch32', d['value'], d['ms_per_step'], d['roofline']['frac'], d['stages_ms_per_step'])"; done
