#!/bin/bash
# the five kernels of the VFE+scatter group INSIDE the frame (serial single-stream frame graph): bash tools/prof_serial.sh <tag>
O=gpurun_out/${1:-ser}
export TMPDIR=/tmp
mkdir -p $O
export HVPR_BEV_STREAMS=1
timeout 280 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_serial -o bench -- python3 bench.py --no-pipeline --no-cpu-baseline --no-extras --no-parity --steps 50 --warmup 5 > $O/stats_serial.log 2>&1
python3 tools/kstats.py $O/stats_serial/bench_kernel_stats.csv 40 | grep "k_vfe\|k_memory_readout\|k1_keys\|k2_scan\|k3_fill\|k_index"
find $O -name "*kernel_trace.csv" -delete
find $O -name "*agent_info.csv" -delete
