python bench.py > gpurun_out/r02_bench1.json 2> gpurun_out/r02_bench1.err; echo "bench rc $?" > gpurun_out/r02_bench1.rc
python -m pytest tests/test_00_multirank_gpu.py -x -q -k "bench" 2>&1 | tail -15 > gpurun_out/r02_bench_tests.txt
