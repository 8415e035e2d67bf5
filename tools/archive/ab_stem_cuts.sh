# How many cuts the stem's filter gradient (4 tiles over 401408 positions) should take: the cut model's slots per CU (tuning build).
P3D_EXTRA_HIPCC_FLAGS=-DP3D_TUNING python -c "
import sys; sys.path.insert(0,'.')
from sap3d_tensorflow_amd import build; build.build(force=True)" > /dev/null 2>&1 || { echo "tuning build failed"; exit 1; }
for sl in 1 2 3 4 6; do echo "slots $sl: $(P3D_WGRAD_SLOTS=$sl timeout -k 10 200 python tools/op_times.py 'stem/conv|deconv4_conv1|results' 2>&1 | grep wgrad | awk '{print $1, $4, $5}' | tr '\n' ' ')"; done
