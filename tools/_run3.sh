python -m pytest tests/test_00_multirank_gpu.py -q -k "watchdog" > gpurun_out/r02_tests_f.txt 2>&1
grep -E "passed|failed" gpurun_out/r02_tests_f.txt
