"""Prints the stream-operation trace of one train step (p3d_debug_schedule) of a small p3d_unet: python tools/sched_dump.py > trace.txt"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sap3d_tensorflow_amd import P3DSession, synthetic

base, blocks, shape = 16, (1, 1, 2), (2, 16, 32, 32)
if len(sys.argv) > 1 and sys.argv[1] == "full":
    base, blocks, shape = 64, (3, 8, 36), (8, 16, 112, 112)
s = P3DSession('unet', batch=shape[0], frames=shape[1], height=shape[2], width=shape[3], base=base, blocks=blocks, seed=1)
s.upload(synthetic.synthetic_clip(0, shape + (3,)), synthetic.synthetic_target(3, shape))
s.train_step_device(0.5, seed=0)
s.synchronize()
for line in s.schedule(0.5, seed=1):
    print(line)
s.close()
