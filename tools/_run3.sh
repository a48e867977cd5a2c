python -m pytest tests/test_00_multirank_gpu.py -q -x -k "c4" > gpurun_out/r02_tests_d.txt 2>&1
python -m pytest tests/test_gpu_parity.py -q -k "row_partition or variants or random_columns" >> gpurun_out/r02_tests_d.txt 2>&1
grep -E "passed|failed" gpurun_out/r02_tests_d.txt
