mkdir -p gpurun_out/r5k
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --kernels > gpurun_out/r5k/bench.json 2> gpurun_out/r5k/kernels_new.txt
P3D_LIB=$PWD/tools/ab/libp3dhip_base.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline --kernels > gpurun_out/r5k/bench_base.json 2> gpurun_out/r5k/kernels_base.txt
echo NEW; grep "igemm2_kernel<64,64>\|bn_small_fwd" gpurun_out/r5k/kernels_new.txt
echo BASE; grep "igemm2_kernel<64,64>\|bn_small_fwd" gpurun_out/r5k/kernels_base.txt
