"""X1 (fp16 storage of the pointwise-conv operands, BASELINE configs[4]) priced from a profiled train step:
time and algorithmic bytes per op kind at 8 x 32x224x224 (or any shape), fp32 against --pointwise fp16, and the share of
the step that halving the stored activations around the 1x1x1 convs could touch.
    python tools/x1_breakdown.py --frames 32 --size 224 --batch 8 > profiles/r03_x1_breakdown.json
Every launch of p3d_profile_step carries its op kind, device time and algorithmic bytes (net.hip, Op::bytes / bbytes)."""
import argparse
import collections
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def profile(a, fp16):
    from sap3d_tensorflow_amd import P3DSession, synthetic as law
    sess = P3DSession(a.structure, batch=a.batch, frames=a.frames, height=a.size, width=a.size, seed=0)
    sess.set_adam(1e-4)
    if fp16:
        sess.set_pointwise_fp16(True)
    shape = (a.batch, a.frames, a.size, a.size)
    sess.upload(law.synthetic_clip(0, shape + (3,)), law.synthetic_target(1, shape))
    for _ in range(2):
        sess.train_step_device(0.5, 1)
    sess.synchronize()
    best = None
    for rep in range(2):
        recs = sess.profile_step(0.5, 2 + rep)
        if best is None:
            best = recs
        else:
            for b, r in zip(best, recs):
                b["ms"] = min(b["ms"], r["ms"])
    sess.close()
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--structure", default="unet")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--size", type=int, default=224)
    a = ap.parse_args()
    out = {"shape": [a.batch, a.frames, a.size, a.size], "structure": a.structure, "modes": {}}
    for fp16 in (False, True):
        recs = profile(a, fp16)
        tot = sum(r["ms"] for r in recs)
        agg = collections.OrderedDict()
        for r in recs:
            # pointwise convs are the launches whose op name ends in a 1x1x1 conv of a bottleneck (conv1 / conv3 / downsample)
            nm = r["name"]
            pw = nm.startswith("block") and nm.rsplit("/", 1)[-1] in ("conv1", "conv3", "proj")
            k = r["kernel"]
            if k.startswith("igemm2"):
                kind = ("1x1x1 conv" if pw else "k x k x k conv") + ": forward / input gradient"
            elif k.startswith("wgrad2"):
                kind = ("1x1x1 conv" if pw else "k x k x k conv") + ": filter gradient"
            elif k.startswith("bn_") or k.startswith("gn_"):
                kind = "BatchNorm passes"
            elif k.startswith("adam"):
                kind = "Adam"
            else:
                kind = "other (pool, pad, loss, bias sums, memsets)"
            e = agg.setdefault(kind, {"launches": 0, "ms": 0.0, "gbytes": 0.0, "gflop": 0.0})
            e["launches"] += 1; e["ms"] += r["ms"]; e["gbytes"] += r["bytes"] / 1e9; e["gflop"] += r["flops"] / 1e9
        for e in agg.values():
            e["share_of_device_time"] = round(e["ms"] / tot, 4)
            e["gbs"] = round(e["gbytes"] / max(e["ms"], 1e-9) * 1e3, 1)
            e["tflops"] = round(e["gflop"] / max(e["ms"], 1e-9), 2)
            e["ms"] = round(e["ms"], 3); e["gbytes"] = round(e["gbytes"], 3); e["gflop"] = round(e["gflop"], 1)
        out["modes"]["pointwise fp16" if fp16 else "fp32"] = {"device_ms_sum": round(tot, 3), "by_kind": agg}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
