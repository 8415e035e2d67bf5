mkdir -p gpurun_out/r3v
b() { timeout -k 10 240 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3v/x.json 2> gpurun_out/r3v/x.err; echo "$1 $(python3 -c "import json;d=json.load(open('gpurun_out/r3v/x.json'));print(d['ms_per_step'])")"; }
rb() { P3D_EXTRA_HIPCC_FLAGS="$1" python -c "
import sys; sys.path.insert(0,'.')
from sap3d_tensorflow_amd import build; build.build(force=True)" > gpurun_out/r3v/build.log 2>&1; echo "rebuild [$1] rc=$?"; }
b base
rb -DP3D_TUNE_NO_LOOP; b igemm2_no_loop
rb -DP3D_TUNE_NO_MFMA; b igemm2_no_mfma
rb -DP3D_TUNE_NO_DMA; b igemm2_no_dma
export P3D_TUNE_SKIP_SIDE=deconv,results,block,stem
rb -DP3D_TUNE_NO_LOOP; b no_loop_and_no_side
