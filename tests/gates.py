"""Measured gradient gates.

The HIP path is bit-reproducible (tests/test_gpu_determinism.py), so for a given test case the distance of its gradients
from the float64 oracle is a fixed number.  Where that number exceeds the fp32-noise bound (5 x the float32 oracle's own
distance + 2e-3) it is because a ReLU / max-pool / arg-max decision of the fp32 forward differs from the float64
oracle's; how much is MEASURED per case and committed in tests/golden/measured_gates.json, and a
case passes when it stays within `margin` x its measured value (+ a small absolute floor).  A case that was measured at 0
therefore has to meet the fp32-noise bound itself: there is no blanket allowance.

Two things keep a re-measurement from absorbing a real regression:
 * CEILINGS below are per case, independent of the measurements and NOT re-measured: what a single flipped decision has been
   seen to cost on that graph (round 2's fixed gates, and round 3's largest measurement x 2 for the cases round 2 did not
   have).  A value above its ceiling fails whatever measured_gates.json says.
 * a measuring run (P3D_MEASURE_GATES) fails a case that grew by more than 1.5 x over the committed value, unless
   P3D_ACCEPT_GATE_GROWTH=1 says the growth was looked at (a new summation order can flip another decision; a kernel bug
   looks the same from here, so the one who re-measures has to say so).

Re-measure after a change of summation order (new tile shape, new fusion):
    gpurun -- 'P3D_MEASURE_GATES=gpurun_out/gates.json python -m pytest tests -m gpu -q'
and copy gpurun_out/gates.json over tests/golden/measured_gates.json.
"""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(HERE, "golden", "measured_gates.json")
MEASURE_TO = os.environ.get("P3D_MEASURE_GATES")          # a path: record instead of asserting
ACCEPT_GROWTH = os.environ.get("P3D_ACCEPT_GATE_GROWTH") == "1"

_data = json.load(open(PATH)) if os.path.exists(PATH) else {}

# per-case ceilings, never re-measured (prefix match, longest prefix wins; DEFAULT for everything else)
DEFAULT_CEILING = 1e-2
CEILINGS = {
    "config2/worst_ratio": 4.0,
    "gn_decoder/": 5e-2,
    "gn_cbam/": 1.5e-2,
    "golden/gn_": 3e-2,
    "golden/unet_b8_333": 2e-2,
    "golden/unetpp_": 1e-2,
    "golden/": 5e-3,
    "unetpp_ds_dropout/": 5e-2,
    "unetpp_ds/": 1.5e-2,
    "unetpp_nonsa/": 1.5e-2,
}


def ceiling(key):
    best = None
    for prefix, value in CEILINGS.items():
        if key.startswith(prefix) and (best is None or len(prefix) > len(best[0])):
            best = (prefix, value)
    return best[1] if best else DEFAULT_CEILING


def _record(key, value):
    cur = json.load(open(MEASURE_TO)) if os.path.exists(MEASURE_TO) else {}
    cur[key] = max(float(value), float(cur.get(key, 0.0)))
    os.makedirs(os.path.dirname(os.path.abspath(MEASURE_TO)), exist_ok=True)
    with open(MEASURE_TO, "w") as f:
        json.dump(cur, f, indent=1, sort_keys=True)


def check(key, value, margin=2.0, floor=1e-4, detail=None):
    """`value` (a non-negative error measure of case `key`) must stay under its fixed ceiling and within margin x the committed
    measurement + floor."""
    assert value <= ceiling(key), ("above the fixed ceiling of this case", key, value, ceiling(key), detail)
    if MEASURE_TO:
        _record(key, value)
        old = _data.get(key)
        if old is not None and not ACCEPT_GROWTH:
            assert value <= 1.5 * old + floor, ("grew by more than 1.5 x while re-measuring: look at it, then P3D_ACCEPT_GATE_GROWTH=1",
                                                key, value, old, detail)
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
