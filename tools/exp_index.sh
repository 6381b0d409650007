#!/bin/bash
# A/B of the index phase of the encode group: three launches (HVPR_INDEX_FUSED=0) against the one-launch kernel, with its hand-offs in
# one XCD's L2 (default) or (HVPR_INDEX_AGENT=1) at device scope; the timing build prints the phase stamps: bash tools/exp_index.sh
O=gpurun_out/exp_index
export TMPDIR=/tmp
mkdir -p $O
for V in "0 0" "1 0" "1 1"; do
    set -- $V
    export HVPR_INDEX_FUSED=$1 HVPR_INDEX_AGENT=$2
    echo "=== HVPR_INDEX_FUSED=$1 HVPR_INDEX_AGENT=$2"
    timeout 300 python3 -m pytest tests/test_gpu_stage1.py -x -q -m gpu -k "encode or voxelize" > $O/test_$1_$2.log 2>&1; echo "tests rc=$?"; tail -2 $O/test_$1_$2.log
    timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p$1_$2 -o g -- python3 tools/bench_group.py --car1 > $O/p$1_$2.log 2>&1
    python3 tools/kstats.py $O/p$1_$2/g_kernel_stats.csv 8 | grep "k_index\|k1\|k2\|k3"
    timeout 100 python3 tools/bench_group.py --car1 | cut -c1-200
    HVPR_AMD_LIB=$PWD/hvpr_amd/libhvpr_amd_timing.so timeout 100 python3 tools/bench_group.py --car1 2>&1 | grep "k_index tile" | tail -3 | cut -c1-330
done
find $O -name "*kernel_trace.csv" -delete
find $O -name "*agent_info.csv" -delete
