#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out/r03j; mkdir -p $OUT
for ht in 1 0; do
  echo "== CTPN_C3_HALFTAIL=$ht"
  CTPN_C3_HALFTAIL=$ht timeout 600 python -m pytest tests -m gpu -q -k "batch_equals or repeatable or production_path or every_layer_at_600" 2>&1 | tail -6
done
