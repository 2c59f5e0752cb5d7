#!/bin/bash
# One GPU call for the JPEG decode path: parity tests, files-in / lines-out throughput, per-kernel profile.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
O=$R/gpurun_out/jpeg
mkdir -p "$O"
timeout 300 python -m pytest tests/test_gpu_jpeg.py -q --tb=short -p no:cacheprovider > "$O/pytest.txt" 2>&1
echo "pytest rc=$?"; tail -n 25 "$O/pytest.txt"
timeout 60 python tools/jpeg_loop.py > "$O/loop.json" 2> "$O/loop.err"; echo "loop rc=$?"; cat "$O/loop.json"
timeout 240 python tools/decode_throughput.py --only-gpu --images 768 --distinct 96 --out "$O/throughput.json" > "$O/throughput.log" 2>&1
echo "throughput rc=$?"; tail -n 30 "$O/throughput.log"
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --stats -d "$O/prof" -o jpeg -- python "$R/tools/jpeg_loop.py" --reps 10 > "$O/rocprof.log" 2>&1
echo "rocprof rc=$?"
ls "$O/prof" 2>/dev/null | head
