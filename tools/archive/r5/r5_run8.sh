mkdir -p gpurun_out/r5h
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --kernels --dump-launches gpurun_out/r5h/launches.csv > gpurun_out/r5h/bench.json 2> gpurun_out/r5h/kernels.txt
grep "bn_" gpurun_out/r5h/kernels.txt
