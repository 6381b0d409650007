#!/bin/bash
# rocprofv3 runs behind profiles/ (run on the GPU box from the repo root: `gpurun -- bash tools/run_profiles.sh r01c`, then
# `python tools/make_profiles.py r01c` here).  Kernel statistics and the two PMC passes are separate runs.
set -u
O=gpurun_out/${1:-r01c}
export TMPDIR=/tmp
mkdir -p $O
T="timeout 280"
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_single -o bench -- python3 bench.py --no-pipeline --no-cpu-baseline --steps 50 --warmup 5 > $O/stats_single.log 2>&1
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 bench.py --no-cpu-baseline --steps 50 --warmup 5 > $O/stats.log 2>&1
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_bf16x6 -o bench -- python3 bench.py --no-pipeline --no-cpu-baseline --conv-precision bf16x6 --steps 30 --warmup 5 --probe-steps 0 > $O/stats_bf16x6.log 2>&1
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_bf16x3 -o bench -- python3 bench.py --no-pipeline --no-cpu-baseline --conv-precision bf16x3 --steps 30 --warmup 5 --probe-steps 0 > $O/stats_bf16x3.log 2>&1
$T rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 bench.py --no-pipeline --steps 10 --warmup 3 --no-cpu-baseline --probe-steps 0 > $O/pmc_fetch.log 2>&1
$T rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 bench.py --no-pipeline --steps 10 --warmup 3 --no-cpu-baseline --probe-steps 0 > $O/pmc_write.log 2>&1
find $O -name "*.csv" | head -30
