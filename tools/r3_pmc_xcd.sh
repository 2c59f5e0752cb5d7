#!/bin/bash
# HBM traffic of the conv kernels with / without XCD-local tile ranges in conv3x3_wr_kernel (separate FETCH_SIZE / WRITE_SIZE passes)
set -u
R=$PWD; OUT=$R/gpurun_out/r03f; mkdir -p $OUT
export TMPDIR=/tmp
(time timeout 900 python -m pytest tests -m gpu -q -x -k "every_layer_at_600 or production_path or batch_equals or weights_in_registers or patch_shapes or edge_columns" 2>&1 | tail -8) > $OUT/pytest.txt 2>&1; tail -4 $OUT/pytest.txt
cd /tmp
for x in 0 1; do
  for grp in FETCH_SIZE WRITE_SIZE; do
    CTPN_C3_WR_XCD=$x rocprofv3 --pmc $grp --kernel-trace -d $OUT/raw -o pmc_${grp}_$x -- python $R/bench.py --steps 3 --warmup 1 --cpu-images 0 --no-other-configs --stage-events off > /dev/null 2> $OUT/pmc_${grp}_$x.err
  done
done
cd $R
python - <<PY
import sqlite3
for x in (0, 1):
    res = {}
    for grp in ("FETCH_SIZE", "WRITE_SIZE"):
        db = sqlite3.connect("$OUT/raw/pmc_%s_%d_results.db" % (grp, x))
        for name, val, cnt in db.execute("select kernel_name, sum(value), count(*) from counters_collection where counter_name = ? group by kernel_name", (grp,)):
            if "conv3x3_wr" in name or "conv3x3_p" in name or "conv_first" in name:
                res.setdefault(name[:70], {})[grp] = (val * 1024 * (2 if grp == "FETCH_SIZE" else 1) / cnt / 1e6, cnt)
    print("XCD ranges =", x)
    for k, v in sorted(res.items()):
        print("  %-72s read %8.1f MB  write %8.1f MB per launch (%d launches)" % (k, v.get("FETCH_SIZE", (0, 0))[0], v.get("WRITE_SIZE", (0, 0))[0], v.get("FETCH_SIZE", (0, 0))[1]))
PY
rm -rf $OUT/raw
