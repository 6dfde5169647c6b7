timeout -k 10 600 python -m pytest tests/test_msg_fused_gpu.py -x -q > gpurun_out/abl_tests.log 2>&1; echo "rc=$?" >> gpurun_out/abl_tests.log
