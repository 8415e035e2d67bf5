"""One HIP runtime per process, whatever the import order (the process used to die at interpreter exit
with `free(): invalid pointer` when libp3dhip was loaded before torch).  Each order runs in a child process: what is
under test is a clean EXIT."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys
sys.path.insert(0, %r)
order = sys.argv[1]
if order == "torch_first":
    import torch
from sap3d_tensorflow_amd import _lib
_lib.lib()
if order == "ours_first":
    import torch
print(order, {k: len(v) for k, v in _lib.mapped_rocm_runtimes().items()})
if len(sys.argv) > 2:
    import numpy as np
    from sap3d_tensorflow_amd import P3DSession
    s = P3DSession("unet", batch=1, frames=16, height=32, width=32, base=16, blocks=(1, 1, 1), seed=1)
    x = np.zeros((1, 16, 32, 32, 3), np.float32); y = np.zeros((1, 16, 32, 32), np.float32)
    print("loss", s.train_step(x, y, 0.0))
    if sys.argv[2] == "close":
        s.close()
print("exiting")
""" % ROOT


def _run(order, *more):
    r = subprocess.run([sys.executable, "-c", CHILD, order] + list(more), capture_output=True, text=True, timeout=900)
    return r


def test_one_runtime_mapped_in_either_order():
    for order in ("ours_first", "torch_first", "ours_only"):
        r = _run(order)
        assert r.returncode == 0, (order, r.stdout[-2000:], r.stderr[-2000:])
        assert "exiting" in r.stdout
        counts = eval(r.stdout.split(order, 1)[1].split("\n")[0])
        assert all(v == 1 for v in counts.values()), (order, counts)


@pytest.mark.gpu
def test_clean_exit_in_either_order():
    for order in ("ours_first", "torch_first", "ours_only"):
        for how in ("close", "noclose"):        # a session left open is closed by the atexit hook
            r = _run(order, how)
            assert r.returncode == 0, (order, how, r.stdout[-2000:], r.stderr[-2000:])
            assert "exiting" in r.stdout and "loss" in r.stdout
            assert "free()" not in r.stderr and "corruption" not in r.stderr, (order, how, r.stderr[-2000:])
