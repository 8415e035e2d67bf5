"""Diagnostic (tuning build): loss and gradients of one structure with / without the K-sliced tail class, per tensor.
   python tests/probes/tail_probe.py dump <path>   (run twice with P3D_TUNE_NO_TAIL=0 / 1), then  compare <a> <b>"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np


def dump(path, structure="gn_p3d", shape=(2, 16, 112, 112)):
    from sap3d_tensorflow_amd import P3DSession, synthetic
    B, T, H, W = shape
    s = P3DSession(structure, batch=B, frames=T, height=H, width=W, seed=3)
    x = synthetic.synthetic_clip(0, (B, T, H, W, 3))
    y = synthetic.synthetic_target(3, (B, T, H, W))
    theta = s.save()
    rng = np.random.default_rng(0)
    for n in sorted(theta):
        if n.endswith('/beta'):
            theta[n] = rng.uniform(-0.2, 0.2, theta[n].shape).astype(np.float32)
    s.load(theta)
    loss, pred = s.backward(x, y, 0.0)
    out = {"__loss": np.float64(loss), "__pred": pred}
    for n, _, tr in s.variables():
        if tr:
            out[n] = s.get_grad(n)
    np.savez(path, **out)
    print("loss", loss)


def compare(a, b):
    A, B = np.load(a), np.load(b)
    print("loss", float(A["__loss"]), float(B["__loss"]), "pred max diff", np.abs(A["__pred"] - B["__pred"]).max())
    rows = []
    for n in A.files:
        if n.startswith("__"):
            continue
        d = np.linalg.norm(A[n].astype(np.float64) - B[n]) / max(np.linalg.norm(A[n]), 1e-30)
        rows.append((d, n, float(np.linalg.norm(A[n]))))
    order = {n: i for i, n in enumerate(A.files)}
    for d, n, g in sorted(rows, key=lambda r: order[r[1]]):
        if d > 1e-4:
            print("  %-50s rel %.3e |g| %.3e" % (n, d, g))
    print("median", np.median([r[0] for r in rows]), "max", max(rows)[:2])


if __name__ == "__main__":
    if sys.argv[1] == "dump":
        dump(sys.argv[2], *(sys.argv[3:4]))
    else:
        compare(sys.argv[2], sys.argv[3])
