#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_bench.sh <outdir> [extra bench.py flags]
# 1) rocprofv3 --kernel-trace --stats of bench.py   2) two counter-only passes (FETCH_SIZE, WRITE_SIZE)   3) plain bench
# Counter passes never combine with tracing (gpurun refuses that mix).  The program follows `--` directly.
set -e
out=$1; shift
export TMPDIR=/tmp
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $out/bench_under_rocprof.json 2> $out/kt.log
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_fetch -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > $out/pmc_write.log 2>&1
find $out -name "*kernel_stats.csv" | head -3
