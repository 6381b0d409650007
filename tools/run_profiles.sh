#!/bin/bash
# rocprofv3 runs behind profiles/ (run on the GPU box from the repo root: `gpurun -- bash tools/run_profiles.sh r05a`, then
# `python tools/make_profiles.py r05a r04` here).  Kernel statistics and the PMC passes are separate runs (a --pmc run carries no
# other trace domain).  Only the summaries leave the box: per-dispatch traces are deleted.
set -u
O=gpurun_out/${1:-r05a}
export TMPDIR=/tmp
mkdir -p $O
T="timeout 280"
B="--no-cpu-baseline --no-extras --no-parity"
# latency mode, the backbone's branches on their own HIP streams (what `single_graph_latency_mode` of the bench line runs)
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_single -o bench -- python3 bench.py --no-pipeline $B --steps 50 --warmup 5 > $O/stats_single.log 2>&1
# SERIAL: the same frame graph with every kernel on ONE stream (HVPR_BEV_STREAMS=1): kernel durations do not overlap, so
# sum(kernel time) <= stage time and per-kernel fractions mean something
export HVPR_BEV_STREAMS=1
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_serial -o bench -- python3 bench.py --no-pipeline $B --steps 50 --warmup 5 > $O/stats_serial.log 2>&1
$T rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 bench.py --no-pipeline --steps 10 --warmup 3 $B --probe-steps 0 > $O/pmc_fetch.log 2>&1
$T rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 bench.py --no-pipeline --steps 10 --warmup 3 $B --probe-steps 0 > $O/pmc_write.log 2>&1
unset HVPR_BEV_STREAMS
# the headline run (frame pipeline)
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 bench.py $B --steps 50 --warmup 5 > $O/stats.log 2>&1
# the VFE+scatter group at batch 16: what bounds k_vfe and k_memory_readout (SQ counters, two passes of <= 8 SQ slots)
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_group16 -o g -- python3 tools/bench_group.py --only16 > $O/stats_group16.log 2>&1
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_group_dense -o g -- python3 tools/bench_group.py --only-dense > $O/stats_group_dense.log 2>&1
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_group1 -o g -- python3 tools/bench_group.py --car1 > $O/stats_group1.log 2>&1
$T rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $O/pmc_sq1 -o s -- python3 tools/bench_group.py --only16 > $O/pmc_sq1.log 2>&1
$T rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_sq2 -o s -- python3 tools/bench_group.py --only16 > $O/pmc_sq2.log 2>&1
# the training step (BASELINE.json configs[2]): one steady-state step = the kernels between the last two optimiser launches
$T rocprofv3 --kernel-trace --stats -d /tmp/prof_train -o t -- python3 tools/bench_train.py --batch 16 --steps 3 --warmup 2 > $O/train.log 2>&1
python3 tools/prof_db.py /tmp/prof_train/t_results.db 80 --step k_fused_adam | sed -n '/one step/,$p' > $O/train_step_kernels.txt
find $O -name "*kernel_trace.csv" -delete
find $O -name "*agent_info.csv" -delete
du -sh $O; find $O -name "*.csv" | head -40
