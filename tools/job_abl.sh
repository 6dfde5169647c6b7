E3_EXP_LIB=l32s STAMPS=1 timeout -k 10 200 python tools/msg_micro.py >> gpurun_out/stamps.txt 2>&1
