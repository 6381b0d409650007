"""Turn the raw rocprofv3 CSVs under gpurun_out/<run> (written by tools/run_profiles.sh on the GPU box) into the committed
summaries under profiles/.   usage: python tools/make_profiles.py <run dir under gpurun_out> <round prefix, e.g. r03>

  profiles/<r>_kernel_stats_single_graph.csv   rocprofv3 --kernel-trace --stats of `bench.py --no-pipeline` (latency mode, branch streams)
  profiles/<r>_kernel_stats_serial.csv         the same with HVPR_BEV_STREAMS=1: every kernel on one stream, durations do not overlap
  profiles/<r>_kernel_stats_pipeline.csv       same of the default `bench.py` (frame pipeline; also holds the latency-mode and probe replays)
  profiles/<r>_kernel_stats_group_b16.csv      the VFE+scatter group alone at hvpr_car batch 16 (tools/bench_group.py --only16)
  profiles/<r>_kernel_stats_group_dense.csv    ... on the dense scene, BASELINE.json configs[4] (--only-dense);  ..._group_b1.csv: hvpr_car batch 1 (--car1)
  profiles/<r>_pmc_sq_group_b16.csv            SQ counters per kernel of that run (two --pmc passes), averaged per dispatch
  profiles/<r>_pmc_hbm_traffic.csv             per-kernel FETCH_SIZE / WRITE_SIZE (two separate --pmc passes), per frame, serial graph
  profiles/<r>_train_step_kernels.txt          one steady-state training step, kernels by total time
  profiles/roofline_traffic.json               the totals bench.py quotes in `roofline.traffic` / `roofline_mfma.traffic`
HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) KiB: FETCH_SIZE is doubled per the gfx950 correction of MI355X_MICROARCH.md (HBM /
rocprofv3 section); WRITE_SIZE is used as reported."""
import csv
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RAW = os.path.join(ROOT, "gpurun_out", sys.argv[1] if len(sys.argv) > 1 else "r05a")
R = sys.argv[2] if len(sys.argv) > 2 else "r05"
OUT = os.path.join(ROOT, "profiles")


def cp(src, dst):
    src = os.path.join(RAW, src)
    if os.path.exists(src):
        shutil.copy(src, os.path.join(OUT, dst))
        return True
    print("missing", src)
    return False


cp("stats_single/bench_kernel_stats.csv", f"{R}_kernel_stats_single_graph.csv")
cp("stats_serial/bench_kernel_stats.csv", f"{R}_kernel_stats_serial.csv")
cp("stats/bench_kernel_stats.csv", f"{R}_kernel_stats_pipeline.csv")
cp("stats_group16/g_kernel_stats.csv", f"{R}_kernel_stats_group_b16.csv")
cp("stats_group_dense/g_kernel_stats.csv", f"{R}_kernel_stats_group_dense.csv")
cp("stats_group1/g_kernel_stats.csv", f"{R}_kernel_stats_group_b1.csv")
cp("train_step_kernels.txt", f"{R}_train_step_kernels.txt")


def per_kernel(path, by_counter=False):
    tot, cnt = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(path)):
        key = (r["Kernel_Name"], r["Counter_Name"]) if by_counter else r["Kernel_Name"]
        tot[key] += float(r["Counter_Value"])
        cnt[key] += 1
    return tot, cnt


# ---- SQ counters of the group at batch 16
sq_rows = {}
for d in ("pmc_sq1", "pmc_sq2"):
    path = os.path.join(RAW, d, "s_counter_collection.csv")
    if not os.path.exists(path):
        print("missing", path)
        continue
    tot, cnt = per_kernel(path, by_counter=True)
    for (k, c), v in tot.items():
        sq_rows.setdefault(k, {})[c] = v / cnt[(k, c)]
if sq_rows:
    counters = sorted({c for v in sq_rows.values() for c in v})
    with open(os.path.join(OUT, f"{R}_pmc_sq_group_b16.csv"), "w") as f:
        f.write("# rocprofv3 --kernel-trace --pmc <SQ counters> (two passes) of: tools/bench_group.py --only16 (hvpr_car batch 16: 262 144 points, ~61 k pillars); average per dispatch\n")
        f.write("# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_BUSY_CYCLES is summed over the SQs (MI355X_MICROARCH.md)\n")
        w = csv.writer(f)
        w.writerow(["kernel"] + counters)
        for k in sorted(sq_rows, key=lambda k: -sq_rows[k].get("SQ_WAVE_CYCLES", 0)):
            w.writerow([k[:100]] + [round(sq_rows[k].get(c, float("nan")), 1) for c in counters])

# ---- HBM traffic per frame (serial frame graph)
f_tot, f_cnt = per_kernel(os.path.join(RAW, "pmc_fetch", "f_counter_collection.csv"))
w_tot, w_cnt = per_kernel(os.path.join(RAW, "pmc_write", "w_counter_collection.csv"))
frames = f_cnt[[k for k in f_cnt if "k_vfe" in k or "k_encode" in k][0]]          # runs exactly once per frame
rows, group, conv, conv_parts = [], 0.0, 0.0, defaultdict(float)
GROUP = ("k_index", "k1_keys", "k2_scan", "k3_fill", "k4_gather", "k_vfe", "k_memory_readout", "k_cell_map", "k_scatter", "k_encode")
CONV = ("k_conv", "k_wino", "k_wgrad", "k_spatial_gate", "k_deconv", "k_head", "k_pool")     # everything of the backbone + head stage
for k in sorted(f_tot, key=lambda k: -(2 * f_tot[k] + w_tot.get(k, 0))):
    calls = f_cnt[k] / frames
    fk, wk = f_tot[k] / f_cnt[k], w_tot.get(k, 0.0) / max(w_cnt.get(k, 1), 1)
    b = (2 * fk + wk) * 1024 * calls
    rows.append((k[:110], round(calls, 2), round(fk, 1), round(wk, 1), int(b)))
    if any(g in k for g in GROUP):
        group += b
    for c in CONV:
        if c in k:
            conv += b
            conv_parts[c] += b
            break
with open(os.path.join(OUT, f"{R}_pmc_hbm_traffic.csv"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --pmc FETCH_SIZE  and  --pmc WRITE_SIZE (two separate passes) of: HVPR_BEV_STREAMS=1 bench.py --no-pipeline --steps 10 --warmup 3 --no-cpu-baseline --no-extras --probe-steps 0\n")
    f.write("# per-frame HBM bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 * calls_per_frame ; the factor 2 is the gfx950 FETCH_SIZE correction (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is used as reported\n")
    w = csv.writer(f)
    w.writerow(["kernel", "calls_per_frame", "FETCH_SIZE_KB_per_call", "WRITE_SIZE_KB_per_call", "hbm_bytes_per_frame"])
    w.writerows(rows)
json.dump({"vfe_scatter_group_bytes": int(group), "conv_stack_bytes": int(conv),
           "conv_stack_bytes_by_kernel": {k: int(v) for k, v in conv_parts.items()},
           "source": f"profiles/{R}_pmc_hbm_traffic.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH_SIZE doubled per the gfx950 "
                     "correction); conv_stack_bytes = every kernel of the backbone + head stage (k_wino*, k_conv*, k_spatial_gate, ...)"},
          open(os.path.join(OUT, "roofline_traffic.json"), "w"), indent=1)
print("frames", frames, "group MB", group / 1e6, "conv GB", conv / 1e9, dict(conv_parts))
for r in rows[:14]:
    print(r)
