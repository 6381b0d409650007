#!/bin/bash
# SQ counters of the Winograd weight-gradient kernel (tools/bench_wgrad.py): bash tools/prof_wgrad_sq.sh <tag>
O=gpurun_out/${1:-wgsq}
export TMPDIR=/tmp
mkdir -p $O
timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $O/p3 -o s -- python3 tools/bench_wgrad.py > $O/p3.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA --output-format csv -d $O/p1 -o s -- python3 tools/bench_wgrad.py > $O/p1.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/p2 -o s -- python3 tools/bench_wgrad.py > $O/p2.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -o s -- python3 tools/bench_wgrad.py > $O/st.log 2>&1
python3 - <<PY
import csv, collections, glob
for d in ("p1","p2","p3"):
    fs = glob.glob("$O/%s/*counter_collection.csv" % d)
    if not fs: print(d, "no csv"); continue
    tot = collections.defaultdict(float); cnt = collections.defaultdict(int)
    for r in csv.DictReader(open(fs[0])):
        k = (r["Kernel_Name"].replace("(anonymous namespace)::","")[:30], r["Grid_Size"], r["Counter_Name"])
        tot[k] += float(r["Counter_Value"]); cnt[k] += 1
    for k in sorted(tot):
        if "k_wgrad_wino(" in k[0] or k[0].startswith("k_wgrad_wino"): print(d, k[0], k[1], k[2], "%.5g" % (tot[k]/cnt[k]), cnt[k])
PY
grep -h "k_wgrad" $O/st/*kernel_stats.csv | cut -c1-200
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
