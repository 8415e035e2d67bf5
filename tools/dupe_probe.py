"""Diagnostic: which ROCm libraries end up loaded twice when libp3dhip is loaded before / after torch."""
import sys, os, re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
order = sys.argv[1]
def load_ours():
    from sap3d_tensorflow_amd import _lib
    _lib.lib()
if order == "ours_first":
    load_ours(); import torch
else:
    import torch; load_ours()
libs = {}
for line in open("/proc/self/maps"):
    m = re.search(r"(/\S+/(lib(amdhip64|rccl|hsa-runtime64|rocblas|hiprtc)[^/\s]*))", line)
    if m: libs.setdefault(m.group(3), set()).add(m.group(1))
for k, v in sorted(libs.items()): print(order, k, sorted(v))
