#!/bin/bash
# Effective clock of every conv kernel under the benchmark's random data and under all-zero data (bench.py --zero-data): GRBM_GUI_ACTIVE per
# launch (cycles) and the launch's duration from the same rocprofv3 pass.  usage: bash tools/r3_clock.sh TAG  -> gpurun_out/TAG/clock.txt
set -u
TAG=${1:-clock}; R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
: > $OUT/clock.txt
for v in random zero; do
  extra=""; [ $v = zero ] && extra="--zero-data"
  timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $OUT/raw_$v -o p -- python $R/bench.py --steps 4 --warmup 1 --cpu-images 0 --no-other-configs --stage-events off $extra > $OUT/bench_$v.json 2> $OUT/err_$v.txt
  echo "== $v data" >> $OUT/clock.txt
  python3 - $OUT/raw_$v >> $OUT/clock.txt 2>&1 <<'PY'
import sqlite3, glob, sys, collections
db = glob.glob(sys.argv[1] + "/**/*results.db", recursive=True)[0]
con = sqlite3.connect(db)
tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
cyc = collections.defaultdict(lambda: [0.0, 0]); mf = collections.defaultdict(float); busy = collections.defaultdict(float)
for k, n, v, c in con.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name"):
    if n == "GRBM_GUI_ACTIVE": cyc[k] = [v, c]
    if n == "SQ_VALU_MFMA_BUSY_CYCLES": mf[k] = v
    if n == "SQ_BUSY_CYCLES": busy[k] = v
t = sorted([x for x in tabs if "kernel_dispatch" in x], key=len)[0]
cols = [r[1] for r in con.execute("pragma table_info(%s)" % t)]
ks = sorted([x for x in tabs if "kernel_symbol" in x], key=len)[0]
kc = [r[1] for r in con.execute("pragma table_info(%s)" % ks)]
idc = "id" if "id" in kc else kc[0]
nmc = "display_name" if "display_name" in kc else ("kernel_name" if "kernel_name" in kc else [c for c in kc if "name" in c][0])
names = dict(con.execute("select %s,%s from %s" % (idc, nmc, ks)))
kid = "kernel_id" if "kernel_id" in cols else [c for c in cols if "kernel" in c][0]
dur = collections.defaultdict(lambda: [0.0, 0])
for k, a, b in con.execute("select %s,start,end from %s" % (kid, t)):
    d = dur[names.get(k, str(k))]; d[0] += b - a; d[1] += 1
tot_c = tot_t = 0.0
for k in sorted(cyc):
    if "conv3x3" not in k or "edge" in k: continue
    dk = [v for n, v in dur.items() if n.split("(")[0] in k or k.split("(")[0] in n]
    if not dk: continue
    ns = dk[0][0] / dk[0][1]; c = cyc[k][0] / cyc[k][1]
    tot_c += cyc[k][0]; tot_t += dk[0][0] * cyc[k][1] / dk[0][1]
    print("%-84s launches %3d  cycles/launch (per XCD) %10.0f  us/launch %7.1f  clock %.3f GHz  mfma/busy %.3f" % (k[:84], cyc[k][1], c / 8, ns / 1e3, c / 8 / ns, mf[k] / max(busy[k], 1)))
print("all conv3x3 main launches: %.3f GHz (GRBM_GUI_ACTIVE is summed over the 8 XCDs)" % (tot_c / 8 / max(tot_t, 1)))
PY
  rm -rf $OUT/raw_$v
done
python3 - <<PY >> $OUT/clock.txt
import json
for v in ("random", "zero"):
    try:
        d = json.load(open("$OUT/bench_%s.json" % v)); print("bench line, %s data: %.1f images/s, conv stack frac %.4f (under the PMC pass)" % (v, d["value"], d["roofline"]["frac"]))
    except Exception as e: print(v, "no bench line", e)
PY
cat $OUT/clock.txt
