#!/bin/bash
# The parity subset of the GPU tests under every kernel-variant switch that selects a different product path (one gpurun call):
#   gpurun --timeout 1500 -- 'bash tools/switch_matrix.sh'
# Each block must end in "passed"; the defaults are covered by the plain `pytest -m gpu` run.
for e in CTPN_TAIL_OVERLAP=1 CTPN_C3_STACK=0 CTPN_C3_HALFTAIL=0 CTPN_C3_AHEAD=0 CTPN_C3_AHEAD=1 CTPN_C3_WR_XCD=0 CTPN_NMS_COLUMNS=0 CTPN_CONV1_MFMA=1 CTPN_C3_PERSIST=0 CTPN_C3_WR=0 CTPN_C3_EDGE=0; do
  echo "== $e"
  env $e timeout 300 python -X faulthandler -m pytest tests -m gpu -q -x -k "every_layer_at_600 or fixtures or odd_shapes_bf16 or production_path or async or batch_equals" 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -4
done
