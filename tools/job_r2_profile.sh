export TMPDIR=/tmp
for v in w1u1p2 w2u1p2 w1u1p1 w2u2p1 w3u2p1 w4u3p1; do E3_EXP_LIB=$v python tools/msg_micro.py 2>&1 | grep "^\["; done > gpurun_out/r2_variants.log 2>&1
timeout -k 10 300 python -m pytest tests/test_msg_fused_gpu.py tests/test_parity_bench_mode_gpu.py -x -q > gpurun_out/r2_t6.log 2>&1; echo "rc=$?" >> gpurun_out/r2_t6.log
tools/profile_bench.sh gpurun_out/r2_prof > gpurun_out/r2_prof.log 2>&1
python tools/traffic_json.py gpurun_out/r2_prof "msg_fused_kernel<2, 2>" gpurun_out/r2_traffic_lmax2.json "tools/profile_bench.sh: rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE / --pmc WRITE_SIZE (separate counter-only passes) -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline" >> gpurun_out/r2_prof.log 2>&1
python tools/traffic_json.py gpurun_out/r2_prof "tp_fwd_mfma_r16_kernel<2, 3, 1, 1, true, 2, false, 0, 1, 2, 0, 1, 2, 0>" gpurun_out/r2_traffic_bf16_lmax2.json "same passes; bf16 leg of bench.py: message TP #1 in bf16 storage" >> gpurun_out/r2_prof.log 2>&1
cp $(find gpurun_out/r2_prof/kt -name "*kernel_stats.csv" | head -1) gpurun_out/r2_bench_kernel_stats.csv
python bench.py > gpurun_out/r2_b6.log 2>&1; echo "bench rc=$?" >> gpurun_out/r2_b6.log
