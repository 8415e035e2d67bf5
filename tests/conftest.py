import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


_abort_trace_installed = False


@pytest.fixture(autouse=True)
def _k_slice_scratch_stays_clean(request):
    """After every GPU test: no arrival counter of the K-slice exchange may be left non-zero (a launch that does would hand a
    premature "last arriver" to whichever launch uses that counter next -- in another session of the same process)."""
    global _abort_trace_installed
    if request.node.get_closest_marker("gpu") is not None and not _abort_trace_installed:
        from sap3d_tensorflow_amd import lib
        lib().p3d_debug_install_abort_trace()       # a C-level backtrace if the process dies inside HIP or the library
        _abort_trace_installed = True
    yield
    if request.node.get_closest_marker("gpu") is None:
        return
    from sap3d_tensorflow_amd import lib
    dirty = lib().p3d_debug_dirty_counters()
    assert dirty == 0, "%d arrival counters left non-zero by %s" % (dirty, request.node.nodeid)
