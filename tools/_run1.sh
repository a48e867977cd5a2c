python -m pytest tests/test_gpu_parity.py -x -q -k "batched or spmm" 2>&1 | tail -3 > gpurun_out/r02_c5_tests.txt
for rw in 1 2 4 8; do for strip in 64 128; do
  python tools/bench_batched.py --rows 1 --strip $strip --rw $rw --steps 6 > gpurun_out/r02_c5f_rw${rw}_s${strip}.json 2>&1
done; done
cat gpurun_out/r02_c5_tests.txt
