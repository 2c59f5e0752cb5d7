set -u
R=$PWD; OUT=$R/gpurun_out/fuse8; mkdir -p $OUT
L=$R/text-detection-ctpn_amd/libctpn_hip_dws.so
timeout 600 python -m pytest tests/test_gpu_round4.py "tests/test_gpu_parity.py::test_batch_equals_singles_and_is_idempotent" -q -x 2>&1 | grep -E "passed|failed|error" | tail -3
for rep in 1 2 3; do
for v in new old; do
  if [ $v = old ]; then export CTPN_LIB_PATH=$L; else unset CTPN_LIB_PATH; fi
  python bench.py --batch 1 --steps 300 --warmup 30 --cpu-images 0 --no-other-configs --stage-events off > $OUT/b1_${v}_$rep.json 2>/dev/null
  python -c "import json; d=json.loads([l for l in open('$OUT/b1_${v}_$rep.json') if l.startswith('{')][0]); print('batch1 $v', d['value'], d['ms_per_step'])"
done; done
for rep in 1 2; do
for v in new old; do
  if [ $v = old ]; then export CTPN_LIB_PATH=$L; else unset CTPN_LIB_PATH; fi
  python bench.py --steps 40 --warmup 10 --cpu-images 0 --no-other-configs > $OUT/b32_${v}_$rep.json 2>/dev/null
  python -c "import json; d=json.loads([l for l in open('$OUT/b32_${v}_$rep.json') if l.startswith('{')][0]); print('batch32 $v', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done; done
