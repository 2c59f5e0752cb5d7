#!/bin/bash
# Where do the cycles of the persistent conv kernel go with its real stores (CTPN_C3_P_ABL=0) and with the same store instructions aimed at one
# hot KiB (5)?  SQ wave-cycle split + effective clock per kernel, ablation library.  usage: bash tools/r3_pmc_epi.sh TAG
set -u
TAG=${1:-epi}; R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
export CTPN_LIB_PATH=$R/text-detection-ctpn_amd/libctpn_hip_ablation.so
[ -f "$CTPN_LIB_PATH" ] || { echo "build it first: make -C text-detection-ctpn_amd/csrc ablation"; exit 1; }
cd /tmp
: > $OUT/epi.txt
for v in 0 5; do
  CTPN_C3_P_ABL=$v timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -d $OUT/raw$v -o p -- python $R/bench.py --steps 3 --warmup 1 --cpu-images 0 --no-other-configs --stage-events off > /dev/null 2> $OUT/err$v.txt
  echo "== CTPN_C3_P_ABL=$v" >> $OUT/epi.txt
  python3 - $OUT/raw$v >> $OUT/epi.txt 2>&1 <<'PY'
import sqlite3, glob, sys, collections
db = glob.glob(sys.argv[1] + "/**/*results.db", recursive=True)[0]
con = sqlite3.connect(db)
acc = collections.defaultdict(dict)
for k, n, v, c in con.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name"):
    acc[k][n] = (v, c)
tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
dur = {}
try:
    for k, d, c in con.execute("select kernel_name, sum(end - start), count(*) from kernels group by kernel_name"):
        dur[k] = (d, c)
except Exception as e:
    print("no kernels view:", e, [t for t in tabs if "kernel" in t][:6])
for k, d in sorted(acc.items()):
    if "conv3x3_p" not in k: continue
    g = lambda n: d.get(n, (0, 1))[0]
    wc = max(g("SQ_WAVE_CYCLES"), 1)
    n = d["SQ_WAVE_CYCLES"][1]
    line = "%-86s n=%d wait_any %.3f issue_stall %.3f (lds %.3f) active %.3f mfma/busy %.3f gui_active/launch %.0f" % (
        k[:86], n, g("SQ_WAIT_ANY") / wc, g("SQ_WAIT_INST_ANY") / wc, g("SQ_WAIT_INST_LDS") / wc, g("SQ_ACTIVE_INST_ANY") / wc,
        g("SQ_VALU_MFMA_BUSY_CYCLES") / max(g("SQ_BUSY_CYCLES"), 1), g("GRBM_GUI_ACTIVE") / n)
    if k in dur: line += " us/launch %.1f clock GHz %.3f" % (dur[k][0] / dur[k][1] / 1e3, g("GRBM_GUI_ACTIVE") / n / (dur[k][0] / dur[k][1]))
    print(line)
PY
  rm -rf $OUT/raw$v
done
cat $OUT/epi.txt
