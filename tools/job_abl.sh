for v in wd2 ud2; do E3_EXP_LIB=$v timeout -k 10 120 python tools/msg_micro.py 2>&1 | grep "ms/launch" >> gpurun_out/stamps.txt; done
E3_EXP_LIB=stamp STAMPS=1 timeout -k 10 200 python tools/msg_micro.py >> gpurun_out/stamps.txt 2>&1
