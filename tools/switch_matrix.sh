#!/bin/bash
# The parity subset of the GPU tests under every per-ctx OPTION that selects a different product path (one gpurun call):
#   gpurun --timeout 1500 -- 'bash tools/switch_matrix.sh'
# Options are ctpn_set_option values of the C ABI (include/ctpn_hip.h); the Python binding maps these environment variables onto the
# option of every Context it creates (_binding.OPTION_ENV) -- the library itself reads none of them. Each block must end in "passed";
# the defaults are covered by the plain `pytest -m gpu` run. (Round 3's kernel A/B switches -- CTPN_C3_*, CTPN_CONV_IMPL,
# CTPN_IGEMM_VARIANT -- were removed in round 4 together with the paths that lost.)
for e in CTPN_TAIL_OVERLAP=1 CTPN_NMS_COLUMNS=0 CTPN_NMS_COLUMNS=2 CTPN_NMS_COLUMNS=3 CTPN_CONV1_FUSE=0 CTPN_CONV1_MFMA=1 CTPN_CONV1_MFMA=0 CTPN_LSTM_SPLIT=1 CTPN_LSTM_SPLIT=0 CTPN_CONNECT_DEVICE=1 CTPN_NMS_PREFIX=0 CTPN_CONV_P64=0 CTPN_TAIL_CONFINE=1 CTPN_SPLIT_EDGE=0; do
  echo "== $e"
  env $e timeout 300 python -X faulthandler -m pytest tests -m gpu -q -x -k "every_layer_at_600 or every_layer_matches or fixtures or odd_shapes_bf16 or production_path or async or batch_equals or lone_image or config5_geometry or tolerance_at_batch_8 or nms_prefix" 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -4
done
