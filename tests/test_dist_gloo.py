"""World-size-2 (and 3: ragged last shard) gloo rehearsal of the row-sharded path on CPU."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("nproc", [2, 3, 4])
def test_row_sharded_algorithm_on_gloo(nproc):
    port = 29600 + nproc
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(HERE, "_dist_worker.py")]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=240, env=env)
    assert out.returncode == 0 and f"DIST_OK {nproc}" in out.stdout, out.stdout[-3000:]
