#!/bin/bash
# L2 (TCC) hit / miss counters per conv kernel for the default configuration and the persistent kernel's store ablations
# (CTPN_C3_P_ABL = 2: epilogue without its stores, 5: the same store instructions aimed at one hot KiB).
# usage (one gpurun call): bash tools/pmc_l2.sh TAG        -> gpurun_out/TAG/l2.txt
set -u
TAG=${1:-l2}; R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
# the timing-only switches below exist only in the -DCTPN_ABLATION build (round 3): make -C text-detection-ctpn_amd/csrc ablation
export CTPN_LIB_PATH=$R/text-detection-ctpn_amd/libctpn_hip_ablation.so
[ -f "$CTPN_LIB_PATH" ] || { echo "build it first: make -C text-detection-ctpn_amd/csrc ablation"; exit 1; }
cd /tmp
: > $OUT/l2.txt
for v in 0 2 5; do
  CTPN_C3_P_ABL=$v timeout 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $OUT/raw$v -o l2 -- python $R/bench.py --steps 3 --warmup 1 --cpu-images 0 --stage-events off > /dev/null 2> $OUT/err$v.txt
  echo "== CTPN_C3_P_ABL=$v" >> $OUT/l2.txt
  python3 - $OUT/raw$v >> $OUT/l2.txt 2>&1 <<'PY'
import sqlite3, glob, sys, collections
db = glob.glob(sys.argv[1] + "/**/*results.db", recursive=True)[0]
con = sqlite3.connect(db)
acc = collections.defaultdict(dict)
for k, n, v, c in con.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name"):
    acc[k][n] = v / max(c, 1)
for k, d in sorted(acc.items()):
    if "conv3x3" not in k: continue
    h, m = d.get("TCC_HIT_sum", 0), d.get("TCC_MISS_sum", 0)
    print("%-74s hit %12.0f miss %12.0f  hit rate %.3f" % (k[:74], h, m, h / max(h + m, 1)))
PY
  rm -rf $OUT/raw$v
done
cat $OUT/l2.txt
