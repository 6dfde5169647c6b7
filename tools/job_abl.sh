timeout -k 10 600 python -m pytest tests/test_msg_fused_gpu.py tests/test_fullsize_gpu.py tests/test_bf16_gpu.py -x -q > gpurun_out/abl_tests.log 2>&1; echo "rc=$?" >> gpurun_out/abl_tests.log
timeout -k 10 200 python tools/msg_micro.py 2>&1 | grep "ms/launch" >> gpurun_out/stamps.txt
LMAX=1 timeout -k 10 120 python tools/msg_micro.py 2>&1 | grep "ms/launch" >> gpurun_out/stamps.txt
LMAX=1 H=16 timeout -k 10 120 python tools/msg_micro.py 2>&1 | grep "ms/launch" >> gpurun_out/stamps.txt
LMAX=2 H=16 timeout -k 10 120 python tools/msg_micro.py 2>&1 | grep "ms/launch" >> gpurun_out/stamps.txt
