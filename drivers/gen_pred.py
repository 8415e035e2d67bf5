#!/usr/bin/env python3
"""Python-3 counterpart of the reference's sliding-window inference (/root/reference/gen_pred.py) on libp3dhip.

For every video: a 16-frame queue slides by ONE frame (gen_pred.py:100-134); the first window emits all 16 maps,
every later window only its last map (gen_pred.py:154-168); frames are normalised like gen_pred.py:117-121
((RGB - [90,102,98]) / 255 after resizing to 112x112) by the fused GPU pass of sap3d_tensorflow_amd.dataflow.  The reference decodes JPEG folders with cv2 and writes
960x1080 JPEGs; cv2 is outside this path, so a video here is a .npy array [F,H,W,3] uint8 RGB and the maps come
back as a float32 array [F,112,112].  Stride-1 windows are batched (`--batch`) instead of run one by one:
`p3d_predict_windows` gives every window the result of its own batch-of-1 run (the backbone BatchNorm uses batch
statistics even at inference, p3d.py:140, so a plain batched forward would couple the windows)."""
import argparse
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def preprocess(video_u8, device=0):
    """gen_pred.py:117-121 per frame (RGB - mean, resize to 112, / 255) as one fused GPU pass (csrc/metrics.hip);
    the kernel takes cv2's BGR order, the .npy videos here are RGB."""
    from sap3d_tensorflow_amd import dataflow
    return dataflow.mapf_frames(np.ascontiguousarray(video_u8[..., ::-1]), 112, device=device)


def predict_video(sess, frames, batch):
    """frames [F,112,112,3] normalised -> saliency [F,112,112] with the reference's write-out rule."""
    F = len(frames)
    if F < 16:
        raise ValueError("need at least 16 frames")
    out = np.zeros((F, 112, 112), np.float32)
    starts = list(range(F - 15))
    for i in range(0, len(starts), batch):
        chunk = starts[i:i + batch]
        clips = np.stack([frames[s:s + 16] for s in chunk] + [frames[chunk[-1]:chunk[-1] + 16]] * (batch - len(chunk)))
        maps = sess.predict_windows(clips)[..., 0]       # = B batch-of-1 forwards (gen_pred.py:151), see include/p3d_hip.h
        for k, s in enumerate(chunk):
            if s == 0:
                out[:16] = maps[k]            # first window: all 16 maps
            else:
                out[s + 15] = maps[k, -1]     # later windows: the newest frame only
    return out


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--model", type=str, default="", help="checkpoint: a directory with a TF `checkpoint` state file (gen_pred.py:57-64), "
                   "a TF bundle prefix, or an .npz keyed by TF variable names")
    p.add_argument("--structure", type=str, default="unet++ds",
                   help="graph to build: unet++ds = p3d_unetplusplus_ds, the buildable form of what gen_pred.py:46 constructs; or unet, "
                        "concat, unet++nonsa, gn_p3d, gn_p3d_concat, gn_p3d_decoder")
    p.add_argument("--videos", type=str, required=True, help="folder with <name>.npy videos [F,H,W,3] uint8 RGB")
    p.add_argument("--out", type=str, default="pred")
    p.add_argument("--batch", type=int, default=8)
    p.add_argument("--gpu", type=str, default="0")
    args = p.parse_args()
    from sap3d_tensorflow_amd import P3DSession
    sess = P3DSession(args.structure, batch=args.batch, device=int(args.gpu), seed=0)
    if args.model:
        sess.restore(args.model)
    os.makedirs(args.out, exist_ok=True)
    for path in sorted(glob.glob(os.path.join(args.videos, "*.npy"))):
        sal = predict_video(sess, preprocess(np.load(path)), args.batch)
        np.save(os.path.join(args.out, os.path.basename(path)), sal)
        print(os.path.basename(path), sal.shape, float(sal.mean()))
    sess.close()


if __name__ == "__main__":
    main()
