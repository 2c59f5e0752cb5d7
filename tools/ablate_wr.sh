#!/bin/bash
# kernel time of the two conv3x3_wr launches under CTPN_C3_WR_VAR variants (measurement only)
set -u
R=$PWD; OUT=$R/gpurun_out/${1:-abl}; mkdir -p $OUT; export TMPDIR=/tmp
# the timing-only switches below exist only in the -DCTPN_ABLATION build (round 3): make -C text-detection-ctpn_amd/csrc ablation
export CTPN_LIB_PATH=$R/text-detection-ctpn_amd/libctpn_hip_ablation.so
[ -f "$CTPN_LIB_PATH" ] || { echo "build it first: make -C text-detection-ctpn_amd/csrc ablation"; exit 1; }
shift
cd /tmp
for v in "$@"; do
  CTPN_C3_WR_VAR=$v timeout 90 rocprofv3 --kernel-trace -d $OUT/raw_$v -o t -- python $R/bench.py --steps 6 --warmup 2 --cpu-images 0 --stage-events off > $OUT/bench_$v.json 2> $OUT/err_$v.txt
  python - <<PY
import sqlite3,glob,collections
db=glob.glob("$OUT/raw_$v/**/*.db",recursive=True)[0]
c=sqlite3.connect(db)
tabs=[r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
t=sorted([x for x in tabs if "kernel_dispatch" in x],key=len)[0]
cols=[r[1] for r in c.execute("pragma table_info(%s)"%t)]
ks=[x for x in tabs if "kernel_symbol" in x][0]
kc=[r[1] for r in c.execute("pragma table_info(%s)"%ks)]
idc="id" if "id" in kc else kc[0]; nmc="kernel_name" if "kernel_name" in kc else [x for x in kc if "name" in x][0]
names=dict(c.execute("select %s,%s from %s"%(idc,nmc,ks)))
kid="kernel_id" if "kernel_id" in cols else [x for x in cols if "kernel" in x][0]
st="start" if "start" in cols else [x for x in cols if "start" in x][0]; en="end" if "end" in cols else [x for x in cols if "end" in x][0]
acc=collections.defaultdict(list)
for k,a,b in c.execute("select %s,%s,%s from %s"%(kid,st,en,t)):
    n=names.get(k,str(k))
    if "conv3x3_wr" in n: acc[n.split("conv3x3_wr_kernel")[1][:28]].append((b-a)/1000.0)
print("var $v:", {k:(round(sum(v)/len(v),1),len(v)) for k,v in acc.items()})
PY
  rm -rf $OUT/raw_$v
done
