timeout -k 10 600 python -m pytest tests/test_scale_gpu.py tests/test_parity_bench_mode_gpu.py tests/test_segnn_gpu.py tests/test_sharding_gpu.py -x -q > gpurun_out/abl_tests.log 2>&1; echo "rc=$?" >> gpurun_out/abl_tests.log
python bench.py --no-cpu-baseline --no-bf16-leg > gpurun_out/abl_bench.log 2>&1
