#!/bin/bash
# per-kernel statistics of the encode group alone (tools/bench_group.py) on the three workloads: bash tools/prof_group.sh <tag>
O=gpurun_out/${1:-grp}
export TMPDIR=/tmp
mkdir -p $O
for w in car1 only16 only-dense; do
    timeout 280 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$w -o g -- python3 tools/bench_group.py --$w > $O/$w.log 2>&1
    grep workload $O/$w.log | cut -c1-220
    python3 tools/kstats.py $O/$w/g_kernel_stats.csv 7 | grep "k_\|k1\|k2\|k3"
done
find $O -name "*kernel_trace.csv" -delete
find $O -name "*agent_info.csv" -delete
