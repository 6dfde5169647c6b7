# round-2 evidence job (run on the GPU box from the repo root): full GPU suite, rocprof summaries, default bench line
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2_final_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_final_tests.log
tools/profile_bench.sh gpurun_out/r2_prof > gpurun_out/r2_prof.log 2>&1
CMD="tools/profile_bench.sh: rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE / --pmc WRITE_SIZE (separate counter-only passes) -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline"
python tools/traffic_json.py gpurun_out/r2_prof "msg_fused_kernel<2, 2, false>" gpurun_out/r2_traffic_lmax2.json "$CMD" >> gpurun_out/r2_prof.log 2>&1
python tools/traffic_json.py gpurun_out/r2_prof "msg_fused_kernel<2, 2, true>" gpurun_out/r2_traffic_bf16_lmax2.json "$CMD (bf16 leg)" >> gpurun_out/r2_prof.log 2>&1
cp $(find gpurun_out/r2_prof/kt -name "*kernel_stats.csv" | head -1) gpurun_out/r2_bench_kernel_stats.csv
python bench.py > gpurun_out/r2_bench_default.log 2>&1; echo "bench rc=$?" >> gpurun_out/r2_bench_default.log
python bench.py --lmax 1 --no-cpu-baseline > gpurun_out/r2_bench_lmax1.log 2>&1; echo "bench rc=$?" >> gpurun_out/r2_bench_lmax1.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r2_smoke.log
