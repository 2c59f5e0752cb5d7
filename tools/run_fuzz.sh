#!/bin/bash
# The decoder fuzzers and the host entry-point fuzzer against the AddressSanitizer build of the host library (no GPU needed):
#   bash tools/run_fuzz.sh [seconds per process, default 60] [processes per fuzzer, default 2]
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
T=${1:-60}
N=${2:-2}
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
make -C "$R/text-detection-ctpn_amd/csrc" asan -j8 > /dev/null
for f in fuzz_jpeg.py fuzz_png.py fuzz_host_entry_points.py; do
  for s in $(seq 1 "$N"); do
    (CTPN_NO_TORCH=1 CTPN_LIB_PATH="$R/text-detection-ctpn_amd/libctpn_hip_asan.so" LD_PRELOAD="$RT" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1 \
      python "$R/tools/$f" "$s" "$T" 2>&1 | grep -E "ERROR|#[0-3] |mutants|calls|Error|assert" | head -8) &
  done
done
wait
