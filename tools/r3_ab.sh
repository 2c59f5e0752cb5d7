#!/bin/bash
# A/B of kernel variants in one GPU call: for every "NAME:ENV=VAL,ENV=VAL" argument (NAME:- = defaults) the default bench (no CPU legs)
# and a per-layer kernel trace.  usage: bash tools/r3_ab.sh TAG ["pytest -k expr"|-] variants...
set -u
TAG=$1; KEXPR=$2; shift 2
R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
if [ "$KEXPR" != "-" ]; then (time timeout 900 python -m pytest tests -m gpu -q -k "$KEXPR" 2>&1 | tail -30) > $OUT/pytest.txt 2>&1; tail -8 $OUT/pytest.txt; fi
for v in "$@"; do
  name=${v%%:*}; envs=${v#*:}
  (
    if [ "$envs" != "-" ]; then IFS=,; for kv in $envs; do export "$kv"; done; unset IFS; fi
    cd $R
    python bench.py --cpu-images 0 --no-other-configs > $OUT/bench_$name.json 2> $OUT/bench_$name.err
    python -c "import json;d=json.load(open('$OUT/bench_$name.json'));print('$name', d['value'], d['ms_per_step'], d['roofline']['frac'], d['stages_ms_per_step']['conv_gemm'], d['stages_ms_per_step']['conv_first'], d['stages_ms_per_step']['gemm'])" || tail -5 $OUT/bench_$name.err
    cd /tmp
    rocprofv3 --kernel-trace --stats -d $OUT/raw_$name -o trace -- python $R/bench.py --steps 6 --warmup 2 --cpu-images 0 --no-other-configs > /dev/null 2> $OUT/trace_$name.err
    cd $R
    python tools/rocprof_layers.py $OUT/raw_$name/trace_results.db $OUT/layers_$name.csv > /dev/null 2>&1
    rm -rf $OUT/raw_$name
  )
done
python - "$OUT" "$@" <<'PY'
import sys, csv, os
out = sys.argv[1]; names = [v.split(":")[0] for v in sys.argv[2:]]
rows = {}
for n in names:
    p = os.path.join(out, "layers_%s.csv" % n)
    if not os.path.exists(p): continue
    for r in csv.DictReader(open(p)):
        rows.setdefault(int(r["pos"]), {})[n] = (r["kernel"], float(r["avg_us"]))
print("pos kernel " + " ".join(names))
for pos in sorted(rows):
    k = next(iter(rows[pos].values()))[0].replace("_ZN4ctpn", "")[:44]
    print(pos, k, " ".join("%8.1f" % rows[pos][n][1] if n in rows[pos] else "     -" for n in names))
tot = {n: sum(rows[p][n][1] for p in rows if n in rows[p] and "conv3x3_" in rows[p][n][0] and "edge" not in rows[p][n][0]) for n in names}
print("conv3x3 main launches sum:", {n: round(v, 1) for n, v in tot.items()})
PY
