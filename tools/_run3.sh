bash tools/profile.sh r02 > gpurun_out/r02_profile.log 2>&1
bash tools/profile_newton.sh r02 1 > gpurun_out/r02_profile_newton.log 2>&1
python tools/bench_newton.py --steps 10 > gpurun_out/r02_newton_final.json 2>&1
python bench.py > gpurun_out/r02_bench2.json 2> gpurun_out/r02_bench2.err
