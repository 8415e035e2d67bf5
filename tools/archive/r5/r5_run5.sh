P3D_LIB=$PWD/tools/ab/libp3dhip_spread.so timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q 2>&1 | tail -2
tools/ab/ab_libs.sh gpurun_out/r5e/b8 3 "product spread" --steps 30 --warmup 8
tools/ab/ab_libs.sh gpurun_out/r5e/b32 1 "product spread" --batch 32 --steps 6 --warmup 2
tools/ab/ab_libs.sh gpurun_out/r5e/s224 1 "product spread" --frames 32 --size 224 --steps 4 --warmup 2
