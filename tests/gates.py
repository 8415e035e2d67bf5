"""Measured gradient gates.

The HIP path is bit-reproducible (tests/test_gpu_determinism.py), so for a given test case the distance of its gradients
from the float64 oracle is a fixed number.  Where that number exceeds the fp32-noise bound (5 x the float32 oracle's own
distance + 2e-3) it is because a ReLU / max-pool / arg-max decision of the fp32 forward differs from the float64
oracle's; how much is MEASURED per case and committed in tests/golden/measured_gates.json, and a
case passes when it stays within `margin` x its measured value (+ a small absolute floor).  A case that was measured at 0
therefore has to meet the fp32-noise bound itself: there is no blanket allowance.

Re-measure after a change of summation order (new tile shape, new fusion):
    gpurun -- 'P3D_MEASURE_GATES=gpurun_out/gates.json python -m pytest tests -m gpu -q'
and copy gpurun_out/gates.json over tests/golden/measured_gates.json.
"""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(HERE, "golden", "measured_gates.json")
MEASURE_TO = os.environ.get("P3D_MEASURE_GATES")          # a path: record instead of asserting

_data = json.load(open(PATH)) if os.path.exists(PATH) else {}


def _record(key, value):
    cur = json.load(open(MEASURE_TO)) if os.path.exists(MEASURE_TO) else {}
    cur[key] = max(float(value), float(cur.get(key, 0.0)))
    os.makedirs(os.path.dirname(os.path.abspath(MEASURE_TO)), exist_ok=True)
    with open(MEASURE_TO, "w") as f:
        json.dump(cur, f, indent=1, sort_keys=True)


def check(key, value, margin=2.0, floor=1e-4, detail=None):
    """`value` (a non-negative error measure of case `key`) must stay within margin x the committed measurement + floor."""
    if MEASURE_TO:
        _record(key, value)
        return
    assert key in _data, "no measured gate for %r: re-measure (tests/gates.py)" % key
    assert value <= margin * _data[key] + floor, (key, value, _data[key], detail)


def noise_excess(e_hip, e_o32):
    """How far the worst tensor lies above the fp32-noise bound 5 * (float32 oracle's error) + 2e-3, and which one."""
    worst, name = 0.0, None
    for n, e in e_hip.items():
        x = e - (5 * e_o32[n] + 2e-3)
        if x > worst:
            worst, name = x, n
    return worst, name


def grad_gate(key, e_hip, e_o32):
    """Every gradient tensor within the fp32-noise bound, plus what was measured for this case."""
    excess, name = noise_excess(e_hip, e_o32)
    check(key, excess, detail=(name, e_hip.get(name), e_o32.get(name)))
