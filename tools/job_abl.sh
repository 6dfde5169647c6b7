# scratch job for A/B runs on the GPU box (development): edit and run with gpurun -- 'bash tools/job_abl.sh'
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/abl_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/abl_smoke.log
timeout -k 10 600 python -m pytest tests/test_msg_fused_gpu.py tests/test_fullsize_gpu.py -x -q > gpurun_out/abl_tests.log 2>&1; echo "rc=$?" >> gpurun_out/abl_tests.log
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/abl_bench.log 2>&1
