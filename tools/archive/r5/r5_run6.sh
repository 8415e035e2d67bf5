P3D_LIB=$PWD/tools/ab/libp3dhip_slicefast.so timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q 2>&1 | tail -2
tools/ab/ab_libs.sh gpurun_out/r5f/b8 3 "product slicefast" --steps 30 --warmup 8
