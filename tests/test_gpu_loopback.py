"""The row-sharded code path with P > 1 ranks on ONE GPU: ranks are host threads, the communicator is the
library's in-process loopback test backend (RCCL refuses two ranks per GPU)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("kind", ["laplace", "banded", "random", "dense", "denseband"])
@pytest.mark.parametrize("nranks", [2, 3, 4])
def test_sharded_path_on_loopback_ranks(nranks, kind):
    out = subprocess.run([sys.executable, os.path.join(HERE, "_loopback_worker.py"), str(nranks), kind],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert out.returncode == 0 and f"LOOPBACK_OK {nranks} {kind}" in out.stdout, out.stdout[-3000:]


@pytest.mark.parametrize("seed,nranks", [(1, 2), (2, 3), (3, 4), (4, 5), (6, 3)])
def test_sharded_path_on_random_banded_matrices(seed, nranks):
    """Random banded matrices (constant / arbitrary values, ragged rows, odd slab boundaries) through the sharded path:
    in-place halo exchange, the wave-sliced kernels on slabs with row_begin > 0, all solvers."""
    out = subprocess.run([sys.executable, os.path.join(HERE, "_loopback_worker.py"), str(nranks), "fuzz%d" % seed],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert out.returncode == 0 and f"LOOPBACK_OK {nranks} fuzz{seed}" in out.stdout, out.stdout[-3000:]
