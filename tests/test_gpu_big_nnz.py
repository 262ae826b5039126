"""tools/big_nnz.py -- the > 2^31-entry run of VERDICT r05 item 4 -- at a size the suite can afford: the same generator, lowering,
sampled-row oracle check of (#>), two bicgstabSteps against the oracle, once with 32-bit and once with forced 64-bit row pointers.  The
full-size run (n = 10 000 100, nnz = 2.2e9) is `python tools/big_nnz.py --full`; its record is profiles/r06_big_nnz.json."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("options", ["", "force_rp64=1"])
def test_big_nnz_tool_small(options):
    cmd = [sys.executable, os.path.join(ROOT, "tools", "big_nnz.py"), "--k", "40", "--s", "3000", "--sample", "512", "--full", "--reps", "3", "--threads", "4"]
    if options:
        cmd += ["--options", options]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["rows"] == 120000 and d["nnz"] == 120000 * 40
    assert d["props"]["rowptr_bits"] == (64 if options else 32)
    assert d["spmv_sampled"]["within_bound"] and d["spmv_full"]["within_bound"]
    assert d["bicgstab_two_steps"]["x_vs_oracle"] <= 1e-9 and d["bicgstab_two_steps"]["recurrence_vs_true"] <= 1e-12
    assert d["lowered_once_s"] > 0 and d["spmv"]["gbps"] > 0
