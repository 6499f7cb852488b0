# Round-2 call 1: GPU tests, driver-style and default bench lines (reference CPU baseline in the same run), then the
# baseline counters of the kernels this round attacks.
mkdir -p gpurun_out/r02
bash scripts/gpu_tests.sh
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02/bench_driver_style.json 2> gpurun_out/r02/bench_driver_style.err; tail -c 1500 gpurun_out/r02/bench_driver_style.json
python bench.py --no-cpu-baseline > gpurun_out/r02/bench_default.json 2> gpurun_out/r02/bench_default.err; tail -c 800 gpurun_out/r02/bench_default.json
PFX=r02a bash scripts/gpu_r2_evidence.sh > gpurun_out/r02/evidence.log 2>&1
bash scripts/gpu_r2_exp1.sh > gpurun_out/r02/exp1.log 2>&1
tail -40 gpurun_out/r02/exp1.log
