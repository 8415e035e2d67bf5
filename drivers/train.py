#!/usr/bin/env python3
"""Python-3 counterpart of the reference trainer (/root/reference/train.py) on libp3dhip.

Same flags (train.py:21-45), same step semantics (train.py:217-218: dropout 0.5, training=True, Adam lr,
Smooth-L1 sum), same periodic eval forward (train.py:225-226) and checkpoint cadence (train.py:266-267).  The
dataset loaders (dataflow.py, tensorpack, cv2) are out of scope: clips come either from
`--data clips.npz` (arrays x [N,16,112,112,3] already normalised like dataflow.py:204-208, y [N,16,112,112]; raw uint8
frames go through sap3d_tensorflow_amd.dataflow.mapf_frames first) or are synthetic with the loader's value law.
Checkpoints are TensorFlow-1.x V2 bundles `model/<info>/p3d_<step>.ckpt.*` with a `checkpoint` state file, keyed by the
TF variable names of train.py:180-185 (trainables + BN moving statistics): the files the reference's Saver writes and
restores (sap3d_tensorflow_amd/tf_checkpoint.py); `--pretrain` takes such a directory, a bundle prefix, or an .npz.
Every `--validiter` steps the validation pass of train.py:243-264 runs: eval forward over the validation clips, CC / SIM /
AUC_Judd of the LAST frame of every clip (GPU kernels, sap3d_tensorflow_amd.metrics), NaNs dropped, means printed."""
import argparse
import datetime
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def get_arguments():
    p = argparse.ArgumentParser(description="P3D saliency trainer (MI355X-native)")
    p.add_argument("--normalization", type=str, default="BN", help="BN -> p3d.py graphs, GN -> gn/p3d_gn.py nets (train.py:37; case-insensitive)")
    p.add_argument("--structure", type=str, default="unet",
                   help="unet | concat | unet++ (train.py:149-154).  unet++ builds p3d_unetplusplus_ds with --SA True (the attention head "
                        "that can be built, p3d.py:340) and p3d_unetplusplus_nonsa with --SA False (p3d.py:401); both can also be named directly")
    p.add_argument("--SA", type=lambda v: str(v).lower() in ("1", "true", "yes"), default=True, help="self attention in the unet++ head (train.py:38)")
    p.add_argument("--net", type=str, default="P3D", help="with --normalization gn: P3D | P3D_CONCAT | P3D_DECODER (gn/train_p3d_gn_dataset.py:30,169-180)")
    p.add_argument("--batch", type=int, default=2)
    p.add_argument("--lr", type=float, default=1e-4)
    p.add_argument("--epoch", type=int, default=1)
    p.add_argument("--gpu", type=str, default="0")
    p.add_argument("--pretrain", type=str, default="", help="checkpoint to resume from (train.py:204-210): a directory with a `checkpoint` "
                   "state file (as the reference takes, ./model/<pretrain>/), a TF bundle prefix, or an .npz")
    p.add_argument("--saveiter", type=int, default=1000)
    p.add_argument("--validiter", type=int, default=1000)
    p.add_argument("--plotiter", type=int, default=1000)
    p.add_argument("--info", type=str, default="run")
    # accepted for command-line compatibility with train.py:29-34; they drive the reference's dataset pipeline, which is
    # out of scope here, except for the clip geometry
    p.add_argument("--trainingprops", type=float, default=0.99)
    p.add_argument("--dataset", type=str, default="svsdndhf1k")
    p.add_argument("--videolength", type=int, default=16, help="frames per clip (train.py:32)")
    p.add_argument("--overlap", type=int, default=15)
    p.add_argument("--imagesize", type=int, nargs=2, default=(112, 112), help="clip height width (train.py:34)")
    p.add_argument("--data", type=str, default="", help="npz with x, y; empty = synthetic clips")
    p.add_argument("--steps", type=int, default=20, help="steps per epoch when synthetic")
    p.add_argument("--validclips", type=int, default=4, help="validation batches per validation pass when synthetic")
    return p.parse_args()


def batches(args, rng):
    from sap3d_tensorflow_amd import synthetic as law
    if args.data:
        d = np.load(args.data)
        x, y = d["x"].astype(np.float32), d["y"].astype(np.float32)
        for e in range(args.epoch):
            order = rng.permutation(len(x))
            for i in range(0, len(x) - args.batch + 1, args.batch):
                idx = order[i:i + args.batch]
                yield x[idx], y[idx]
    else:
        for s in range(args.epoch * args.steps):
            shape = (args.batch, args.videolength, args.imagesize[0], args.imagesize[1])
            yield law.synthetic_clip(s, shape + (3,)), law.synthetic_target(10_000 + s, shape)


def validation_batches(args):
    """gt_df of train.py:110-121: the held-out clips (the last 1 - trainingprops share of --data), or synthetic ones."""
    from sap3d_tensorflow_amd import synthetic as law
    if args.data:
        d = np.load(args.data)
        x, y = d["x"].astype(np.float32), d["y"].astype(np.float32)
        first = min(int(len(x) * args.trainingprops), len(x) - args.batch)
        for i in range(max(first, 0), len(x) - args.batch + 1, args.batch):
            yield x[i:i + args.batch], y[i:i + args.batch]
    else:
        for s in range(args.validclips):
            shape = (args.batch, args.videolength, args.imagesize[0], args.imagesize[1])
            yield law.synthetic_clip(500_000 + s, shape + (3,)), law.synthetic_target(600_000 + s, shape)


def validate(sess, args, step):
    """train.py:243-264: CC, SIM, AUC_Judd between the last predicted frame and the last ground-truth frame of every
    validation clip; NaNs (no fixation, flat map) are dropped before averaging."""
    from sap3d_tensorflow_amd import metrics
    print("Doing validation...")
    preds, gts = [], []
    for xs, ys in validation_batches(args):
        image0 = sess.forward(xs, dropout=0.0, training=False)[..., 0]              # train.py:250-251
        preds.append(image0[:, -1]); gts.append(ys[:, -1])                          # prediction[-1], ground_truth[-1]
    if not preds:
        return None
    p, g = np.concatenate(preds), np.concatenate(gts)
    cc, sim, auc = metrics.CC_batch(p, g), metrics.SIM_batch(p, g), metrics.AUC_Judd_batch(p, g)       # jitter on, as train.py:260
    res = tuple(float(np.mean(v[~np.isnan(v)])) if np.any(~np.isnan(v)) else float("nan") for v in (cc, sim, auc))
    print(datetime.datetime.now().isoformat()[:-7], " Step:", step, " Metrics:", *res)
    return res


def main():
    args = get_arguments()
    from sap3d_tensorflow_amd import P3DSession
    gn_nets = {"P3D": "gn_p3d", "P3D_CONCAT": "gn_p3d_concat", "P3D_DECODER": "gn_p3d_decoder"}     # gn/train_p3d_gn_dataset.py:169-180
    gn = args.normalization.lower() == "gn"
    if gn and args.net not in gn_nets:
        raise SystemExit("--net %s is not built (its attention() calls do not match utils/network.py:157 and cannot build "
                         "in the reference either); have %s" % (args.net, sorted(gn_nets)))
    structure = args.structure
    if structure == "unet++":                                                        # train.py:153-154
        structure = "unet++ds" if args.SA else "unet++nonsa"
    if gn:
        structure = gn_nets[args.net]
    sess = P3DSession(structure, batch=args.batch, frames=args.videolength, height=args.imagesize[0], width=args.imagesize[1],
                      device=int(args.gpu), seed=0)                                  # graph + global_variables_initializer
    sess.set_adam(args.lr)
    model_dir = os.path.join("model", args.info)
    os.makedirs(model_dir, exist_ok=True)
    if args.pretrain:
        print(args.pretrain, "Using this model to retrain...")
        sess.restore(args.pretrain)                                                 # train.py:204-210
    print("Start training")
    step = 0
    for xs, ys in batches(args, np.random.default_rng(0)):
        step += 1
        loss = sess.train_step(xs, ys, dropout=0.5, seed=step)                      # train.py:217-218
        if step < 10 or step % args.plotiter == 0:
            image = sess.forward(xs, dropout=0.0, training=False)                   # train.py:225-226
            print("Datetime", datetime.datetime.now().isoformat()[:-7], "Training step:", step,
                  float(np.sum(image[0, -1]) * 255.0), float(np.sum(ys[0][-1]) * 255.0), "Training Loss", loss)
        if step % args.validiter == 0:
            validate(sess, args, step)                                              # train.py:243-264
        if step % args.saveiter == 0:
            sess.save_checkpoint(model_dir, step, keep=10)                          # train.py:180-185,266-267
    print("Training Finished!")
    sess.close()


if __name__ == "__main__":
    main()
