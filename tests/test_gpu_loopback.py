"""The row-sharded code path with P > 1 ranks on ONE GPU: ranks are host threads, the communicator is the
library's in-process loopback test backend (RCCL refuses two ranks per GPU)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("kind", ["laplace", "banded", "random"])
@pytest.mark.parametrize("nranks", [2, 3, 4])
def test_sharded_path_on_loopback_ranks(nranks, kind):
    out = subprocess.run([sys.executable, os.path.join(HERE, "_loopback_worker.py"), str(nranks), kind],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert out.returncode == 0 and f"LOOPBACK_OK {nranks} {kind}" in out.stdout, out.stdout[-3000:]
