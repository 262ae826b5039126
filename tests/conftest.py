import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "sparse-linear-algebra_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


# Every GPU test (and every worker process it spawns) runs under the library's device-binding assertion: launches, copies,
# collectives and allocations must come from a thread that is inside an entry point bound to the context they belong to
# (sla_internal.hpp: Bind / stream_of / dev_malloc).  On a one-GPU box a wrong current device is invisible -- every id is 0 --
# so this is how the multi-device fan-out (sla_ctx_create_multi's worker threads) is checked there.  Read when the library loads.
os.environ.setdefault("SLA_DEBUG_BINDING", "1")
# Round 6 made ONE persistent on-chip launch the default way to run bicgstabSteps on small constant-coefficient stencils -- exactly the
# matrices most of this suite uses to exercise the launch flow's kernels (K1 / K23 / K45 with every fused epilogue, the plane march, the
# gather kernel's folded K2 ...), which remain what runs at 10 M rows.  The suite therefore keeps the launch flow as ITS default (the
# environment is a context's default, sla_ctx_set_option overrides it): tests of the on-chip path (test_gpu_onchip.py, the bench contract)
# switch it on explicitly, through the typed option or the child's environment.
os.environ.setdefault("SLA_ONCHIP", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _no_binding_violations(request):
    yield
    if "gpu" in request.keywords and _has_gpu():
        import sla_amd
        assert sla_amd.Context.binding_violations() == 0, "device work issued by a thread not bound to its context (SLA_DEBUG_BINDING)"


def _has_gpu():
    return os.path.exists("/dev/kfd") and os.path.isdir("/dev/dri")


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
