"""Turn the raw rocprofv3 CSVs under gpurun_out/<run> (written by tools/run_profiles.sh on the GPU box) into the committed
summaries under profiles/.

  profiles/r02_kernel_stats_single_graph.csv   rocprofv3 --kernel-trace --stats of `bench.py --no-pipeline` (serial frame graph)
  profiles/r02_kernel_stats_pipeline.csv       same of the default `bench.py` (frame pipeline; also holds the latency-mode and probe replays)
  profiles/r02_kernel_stats_single_graph_bf16x6.csv / _bf16x3.csv   same with --conv-precision bf16x6 / bf16x3
  profiles/r02_pmc_hbm_traffic.csv             per-kernel FETCH_SIZE / WRITE_SIZE (two separate --pmc passes), per frame
  profiles/roofline_traffic.json               the two totals bench.py quotes in `roofline.traffic`
HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) KiB: FETCH_SIZE is doubled per the gfx950 correction of MI355X_MICROARCH.md (HBM /
rocprofv3 section); WRITE_SIZE is used as reported."""
import csv
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RAW = os.path.join(ROOT, "gpurun_out", sys.argv[1] if len(sys.argv) > 1 else "r02a")
OUT = os.path.join(ROOT, "profiles")
FRAMES_PMC = 13 + 6            # bench.py --steps 10 --warmup 3, plus GraphedForward's 3 eager warm-ups and ... (calls are normalised per kernel below)

shutil.copy(os.path.join(RAW, "stats_single", "bench_kernel_stats.csv"), os.path.join(OUT, "r02_kernel_stats_single_graph.csv"))
shutil.copy(os.path.join(RAW, "stats", "bench_kernel_stats.csv"), os.path.join(OUT, "r02_kernel_stats_pipeline.csv"))
src = os.path.join(RAW, "train_step_kernels.txt")
if os.path.exists(src):
    shutil.copy(src, os.path.join(OUT, "r02_train_step_kernels.txt"))
for prec in ("bf16x6", "bf16x3"):
    src = os.path.join(RAW, "stats_" + prec, "bench_kernel_stats.csv")
    if os.path.exists(src):
        shutil.copy(src, os.path.join(OUT, "r02_kernel_stats_single_graph_%s.csv" % prec))


def per_kernel(path):
    tot, cnt = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(path)):
        tot[r["Kernel_Name"]] += float(r["Counter_Value"])
        cnt[r["Kernel_Name"]] += 1
    return tot, cnt


f_tot, f_cnt = per_kernel(os.path.join(RAW, "pmc_fetch", "f_counter_collection.csv"))
w_tot, w_cnt = per_kernel(os.path.join(RAW, "pmc_write", "w_counter_collection.csv"))
frames = f_cnt[[k for k in f_cnt if "k_vfe" in k][0]]          # k_vfe runs exactly once per frame
rows, group, conv = [], 0.0, 0.0
GROUP = ("k1_keys", "k2_scan", "k3_fill", "k4_gather", "k_vfe", "k_memory_readout", "k_cell_map", "k_scatter")   # fused path: 5 of them
for k in sorted(f_tot, key=lambda k: -(2 * f_tot[k] + w_tot.get(k, 0))):
    calls = f_cnt[k] / frames
    fk, wk = f_tot[k] / f_cnt[k], w_tot.get(k, 0.0) / max(w_cnt.get(k, 1), 1)
    b = (2 * fk + wk) * 1024 * calls
    rows.append((k[:110], round(calls, 2), round(fk, 1), round(wk, 1), int(b)))
    if any(g in k for g in GROUP):
        group += b
    if "k_conv" in k:
        conv += b
with open(os.path.join(OUT, "r02_pmc_hbm_traffic.csv"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --pmc FETCH_SIZE  and  --pmc WRITE_SIZE (two separate passes) of: bench.py --no-pipeline --steps 10 --warmup 3 --no-cpu-baseline --probe-steps 0\n")
    f.write("# per-frame HBM bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 * calls_per_frame ; the factor 2 is the gfx950 FETCH_SIZE correction (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is used as reported\n")
    w = csv.writer(f)
    w.writerow(["kernel", "calls_per_frame", "FETCH_SIZE_KB_per_call", "WRITE_SIZE_KB_per_call", "hbm_bytes_per_frame"])
    w.writerows(rows)
json.dump({"vfe_scatter_group_bytes": int(group), "conv_stack_bytes": int(conv),
           "source": "profiles/r02_pmc_hbm_traffic.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH_SIZE doubled per the gfx950 correction)"},
          open(os.path.join(OUT, "roofline_traffic.json"), "w"), indent=1)
print("frames", frames, "group MB", group / 1e6, "conv GB", conv / 1e9)
for r in rows[:14]:
    print(r)
