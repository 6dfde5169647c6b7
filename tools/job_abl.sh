# scratch job for A/B runs on the GPU box (development): edit and run with gpurun -- 'bash tools/job_abl.sh'
timeout -k 10 600 python -m pytest tests/test_msg_fused_gpu.py tests/test_fullsize_gpu.py tests/test_bf16_gpu.py tests/test_parity_bench_mode_gpu.py -x -q > gpurun_out/abl_tests.log 2>&1; echo "rc=$?" >> gpurun_out/abl_tests.log
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --timing-json gpurun_out/abl_timing.json > gpurun_out/abl_bench.log 2>&1
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-bf16-leg > gpurun_out/abl_bench2.log 2>&1
