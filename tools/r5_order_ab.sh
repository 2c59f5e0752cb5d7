#!/bin/bash
# VERDICT r4 item 7, measured: tile order of conv3x3_p_kernel on the Co = 512 layers -- default (an XCD's workers take all four channel
# slices of 8 pixel tiles per round), one slice per XCD (CTPN_C3_P_ABL=256), two slices per XCD (512) -- in the ablation library
# (`make -C text-detection-ctpn_amd/csrc ablation`; results stay correct: the orders are bijections). Per order: the parity subset, images/s of
# the default bench (same box, same call), and FETCH_SIZE per conv kernel (x 2 x 1024 = bytes read through the L2s' misses).
#   gpurun --timeout 1500 -- 'bash tools/r5_order_ab.sh r5order'
TAG=${1:-order}; R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
export CTPN_LIB_PATH=$R/text-detection-ctpn_amd/libctpn_hip_ablation.so
for o in 0 256 512; do
  echo "== CTPN_C3_P_ABL=$o"
  CTPN_C3_P_ABL=$o python -m pytest tests/test_gpu_parity.py -q -x -k "bf16_every_layer_at_600 or batch_equals" 2>&1 | tail -1
  for rep in 1 2; do
    CTPN_C3_P_ABL=$o python bench.py --no-other-configs --cpu-images 0 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('images/s', d['value'], 'frac', d['roofline']['frac'])"
  done
  (cd /tmp && CTPN_C3_P_ABL=$o rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/raw$o -o pmc -- python $R/bench.py --steps 4 --warmup 1 --cpu-images 0 --no-other-configs --stage-events off > /dev/null 2> $OUT/pmc$o.err)
  python - $OUT/raw$o/pmc_results.db <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tot = 0
for name, val, cnt in db.execute("select kernel_name, sum(value), count(*) from counters_collection where counter_name = 'FETCH_SIZE' group by kernel_name"):
    if "conv3x3_p_kernel" in name:
        mb = 2 * val * 1024 / cnt / 1e6
        tot += 2 * val * 1024 / 5
        print("  %-96s launches %3d  read %7.1f MB per launch" % (name[:96], cnt, mb))
print("  conv3x3_p_kernel reads per step: %.2f GB" % (tot / 1e9))
PY
  rm -rf $OUT/raw$o
done 2>&1 | tee $OUT/order_ab.txt
