set -u
R=$PWD; OUT=$R/gpurun_out/fuse5; mkdir -p $OUT
(time timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -12) > $OUT/pytest_gpu.txt 2>&1
tail -6 $OUT/pytest_gpu.txt
python bench.py --batch 1 --steps 200 --warmup 20 --cpu-images 0 --no-other-configs --stage-events off > $OUT/b1.json 2>/dev/null
python -c "import json; d=json.loads([l for l in open('$OUT/b1.json') if l.startswith('{')][0]); print('batch1', d['value'], d['ms_per_step'])"
bash tools/r4_b1.sh gpurun_out/fuse5/b1trace
head -45 gpurun_out/fuse5/b1trace/timeline.txt
