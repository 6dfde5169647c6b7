timeout -k 10 600 python -m pytest tests/test_msg_fused_gpu.py tests/test_fullsize_gpu.py tests/test_bf16_gpu.py tests/test_sharding_gpu.py -x -q > gpurun_out/abl_tests.log 2>&1; echo "rc=$?" >> gpurun_out/abl_tests.log
for i in 1 2; do timeout -k 10 120 python tools/msg_micro.py 2>&1 | grep "ms/launch" >> gpurun_out/stamps.txt; done
LMAX=1 timeout -k 10 120 python tools/msg_micro.py 2>&1 | grep "ms/launch" >> gpurun_out/stamps.txt
