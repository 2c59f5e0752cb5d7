"""Static check of the compiled conv kernels (CPU: hipcc cross-compiles): no LDS read is in flight across an s_barrier.

Round 6 found the persistent conv kernels' only race there (profiles/r06_barrier_war.txt): hipcc sinks register-only work -- MFMAs and the
s_waitcnt for their LDS operands -- below a raw s_barrier, and the LDS-DMA issued right behind that barrier recycled the weight strip those
reads were still aimed at. The kernels now wait lgkmcnt(0) in front of every barrier of the K loop; this test keeps a compiler upgrade or an
edit from quietly undoing that."""
import os
import subprocess
import sys

import pytest


@pytest.mark.parametrize("src", ["conv3x3_bf16.hip", "conv3x3_split.hip", "conv3x3_f16.hip", "conv3x3_f32.hip"])
def test_no_lds_read_in_flight_across_a_barrier(root, src):
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "scan_barrier_reads.py"), src], capture_output=True, text=True, timeout=1200)
    rows = [l for l in r.stdout.splitlines() if "barriers" in l]
    assert rows, r.stdout + r.stderr
    p_rows = [l for l in rows if "conv3x3_p_kernel" in l or "ctpn::conv3x3_kernel" in l]
    assert len(p_rows) >= 10, rows
    assert r.returncode == 0, "\n".join(l for l in rows if "MUST BE CLEAN" in l)
    assert all(l.split()[-1] == "0" for l in p_rows), p_rows
