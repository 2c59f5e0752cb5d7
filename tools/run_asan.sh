#!/bin/bash
# Host code of libctpn_hip.so under AddressSanitizer (the reference has no sanitizer run at all).
#   make -C text-detection-ctpn_amd/csrc asan && bash tools/run_asan.sh [pytest args]
# Runs the CPU tests that exercise the library's host side (ABI, host connector, result writers, resize arithmetic, worker pool,
# thread budget, JPEG parser / entropy decoder and PNG decoder incl. damaged files) against ../libctpn_hip_asan.so. Python itself is not instrumented, so the ASan runtime is preloaded; leak
# detection is off (the interpreter never frees its arenas). Needs no GPU.
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
[ -f "$R/text-detection-ctpn_amd/libctpn_hip_asan.so" ] || make -C "$R/text-detection-ctpn_amd/csrc" asan -j8
cd "$R"
CTPN_NO_TORCH=1 CTPN_LIB_PATH="$R/text-detection-ctpn_amd/libctpn_hip_asan.so" LD_PRELOAD="$RT" \
  ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1 \
  python -m pytest tests/test_abi.py tests/test_host_logic.py tests/test_properties.py tests/test_jpeg.py tests/test_png.py -q -m "not gpu" -p no:cacheprovider "$@"
