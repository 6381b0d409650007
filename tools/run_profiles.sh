#!/bin/bash
# rocprofv3 runs behind profiles/ (run on the GPU box from the repo root: `gpurun -- bash tools/run_profiles.sh r02a`, then
# `python tools/make_profiles.py r02a` here).  Kernel statistics and the two PMC passes are separate runs (a --pmc run carries
# no other trace domain).  Only the summaries leave the box: per-dispatch traces are deleted.
set -u
O=gpurun_out/${1:-r02a}
export TMPDIR=/tmp
mkdir -p $O
T="timeout 280"
B="--no-cpu-baseline --no-extras"
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_single -o bench -- python3 bench.py --no-pipeline $B --steps 50 --warmup 5 > $O/stats_single.log 2>&1
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 bench.py $B --steps 50 --warmup 5 > $O/stats.log 2>&1
$T rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 bench.py --no-pipeline --steps 10 --warmup 3 $B --probe-steps 0 > $O/pmc_fetch.log 2>&1
$T rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 bench.py --no-pipeline --steps 10 --warmup 3 $B --probe-steps 0 > $O/pmc_write.log 2>&1
# the training step (BASELINE.json configs[2]): one steady-state step = the kernels between the last two optimiser launches
$T rocprofv3 --kernel-trace --stats -d /tmp/prof_train -o t -- python3 tools/bench_train.py --batch 16 --steps 3 --warmup 2 > $O/train.log 2>&1
python3 tools/prof_db.py /tmp/prof_train/t_results.db 80 --step k_fused_adam | sed -n '/one step/,$p' > $O/train_step_kernels.txt
find $O -name "*kernel_trace.csv" -delete
find $O -name "*agent_info.csv" -delete
du -sh $O; find $O -name "*.csv" | head -30
