timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/abl_tests.log 2>&1; echo "rc=$?" >> gpurun_out/abl_tests.log
python bench.py --no-cpu-baseline > gpurun_out/abl_bench.log 2>&1
LMAX=1 timeout -k 10 120 python tools/msg_micro.py 2>&1 | grep "ms/launch" >> gpurun_out/stamps.txt
