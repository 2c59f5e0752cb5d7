#!/bin/bash
# VERDICT r4 item 4, measured: what conv2_1 (conv3x3_wr_kernel, the only layer that stores a full-resolution 128-channel map; MFMA busy 0.57 -
# 0.59) spends its CYCLES on. Ablation library, CTPN_C3_WR_VAR: 0 = the product kernel, 1 = no window DMA after the prologue, 2 = no epilogue
# (no stores), 3 = neither. CYCLES (GRBM_GUI_ACTIVE / 8 XCDs per launch), not time: an ablated layer feeds zeros downstream and the part clocks
# up (round 3). usage: gpurun --timeout 900 -- 'bash tools/r5_wr_cycles.sh r5wr'
set -u
TAG=${1:-wr}; R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
export CTPN_LIB_PATH=$R/text-detection-ctpn_amd/libctpn_hip_ablation.so
[ -f "$CTPN_LIB_PATH" ] || { echo "build it first: make -C text-detection-ctpn_amd/csrc ablation"; exit 1; }
cd /tmp
for v in 0 1 2 3; do
  CTPN_C3_WR_VAR=$v timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -d $OUT/raw$v -o p -- python $R/bench.py --steps 3 --warmup 1 --cpu-images 0 --no-other-configs --stage-events off > /dev/null 2> $OUT/err$v.txt
  echo "== CTPN_C3_WR_VAR=$v"
  python3 - $OUT/raw$v <<'PY'
import sqlite3, glob, sys, collections
db = glob.glob(sys.argv[1] + "/**/*results.db", recursive=True)[0]
con = sqlite3.connect(db)
acc = collections.defaultdict(dict)
for k, n, v, c in con.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name"):
    acc[k][n] = (v, c)
for k, d in sorted(acc.items()):
    if "conv3x3_wr" not in k: continue
    g = lambda n: d.get(n, (0, 1))[0]
    n = d.get("GRBM_GUI_ACTIVE", (0, 1))[1]
    wc = max(g("SQ_WAVE_CYCLES"), 1)
    cyc = g("GRBM_GUI_ACTIVE") / 8.0 / max(n, 1)
    print("  %-70s launches %2d  cycles/launch %9.0f  mfma busy %.3f  wave cycles: issue stall %.2f, waitcnt/barrier %.2f, active %.2f"
          % (k.split("conv3x3_wr_kernel")[1][:70], n, cyc, g("SQ_VALU_MFMA_BUSY_CYCLES") / max(g("GRBM_GUI_ACTIVE") / 8.0 * 1024.0, 1), g("SQ_WAIT_INST_ANY") / wc, g("SQ_WAIT_ANY") / wc, g("SQ_ACTIVE_INST_ANY") / wc))
PY
  rm -rf $OUT/raw$v
done 2>&1 | tee $OUT/wr_cycles.txt
