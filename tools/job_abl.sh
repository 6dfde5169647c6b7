# scratch job for A/B runs on the GPU box (development): edit and run with gpurun -- 'bash tools/job_abl.sh'
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/abl_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/abl_smoke.log
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/abl_bench.log 2>&1
