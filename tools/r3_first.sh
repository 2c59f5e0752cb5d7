#!/bin/bash
# fresh-process repeats of tools/first_run_check.py under the given env settings: bash tools/r3_first.sh TAG n reps ENV=VAL...
TAG=$1; N=$2; REPS=$3; shift 3
OUT=gpurun_out/$TAG; mkdir -p $OUT
for e in "$@"; do
  for i in $(seq 1 $REPS); do
    echo "== $e rep $i"
    env $e timeout 200 python tools/first_run_check.py $N 3 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | grep "DIFFERS\|singles\|rror" 
  done
done 2>&1 | tee $OUT/first.txt
