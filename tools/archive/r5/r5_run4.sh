tools/ab/ab_libs.sh gpurun_out/r5d/b8 2 "product nload2 nload4" --steps 30 --warmup 8
tools/ab/ab_libs.sh gpurun_out/r5d/b32 1 "product nload4" --batch 32 --steps 6 --warmup 2
tools/ab/ab_libs.sh gpurun_out/r5d/s224 1 "product nload4" --frames 32 --size 224 --steps 4 --warmup 2
P3D_LIB=$PWD/tools/ab/libp3dhip_nload4.so timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q 2>&1 | tail -2
