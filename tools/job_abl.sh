E3_EXP_LIB=w2u2 timeout -k 10 200 python tools/msg_micro.py 2>&1 | grep "ms/launch" >> gpurun_out/stamps.txt
timeout -k 10 200 python tools/msg_micro.py 2>&1 | grep "ms/launch" >> gpurun_out/stamps.txt
