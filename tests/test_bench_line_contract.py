"""CPU: the committed bench line of the round (profiles/r*_bench_line.json, the stdout of `python bench.py` on an MI355X) carries every
field of the driver's contract — and the numbers inside it are consistent with each other."""
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _latest():
    paths = glob.glob(os.path.join(ROOT, "profiles", "r*_bench_line.json"))
    assert paths
    return max(paths, key=lambda p: int(re.match(r"r(\d+)_", os.path.basename(p)).group(1)))


def test_bench_line_has_the_contract_fields_and_is_self_consistent():
    r = json.load(open(_latest()))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in r, k
    assert r["n_gpus"] == 1 and r["higher_is_better"] is True and r["scaling"] == "weak" and r["vs_baseline"] is None
    assert r["dtype"] == "f32" and r["data"] == "synthetic" and "workload" in r["config"] and "model" not in r["config"]
    assert "hvpr_car" in r["config"]["workload"] and "batch=1" in r["config"]["workload"]
    # value = whole-job frames/s = n_gpus * steps / time
    assert abs(r["value"] - 1e3 / r["ms_per_step"]) < 0.01 * r["value"]
    ro = r["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in ro, k
    assert ro["bound"] == "hbm" and ro["unit"] == "GB/s" and ro["peak"] == 8000.0
    assert abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-4
    # achieved = algorithmic bytes (16 N + 4 * 160 * nx * ny + W) / the measured duration
    assert ro["algorithmic_bytes"] == 16 * 16384 + 4 * 160 * 296 * 248 + 523712
    assert abs(ro["achieved"] - ro["algorithmic_bytes"] / (ro["avg_duration_us"] * 1e-6) / 1e9) < 0.01 * ro["achieved"]
    # the committed profile of the same code agrees with the live figure (bench.py flags it otherwise)
    prof = ro["in_frame_profile"]
    assert prof["consistent_with_live_within_25pct"] is True
    assert os.path.exists(os.path.join(ROOT, "profiles", prof["file"]))
    cb = r["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0
    mf = r["roofline_mfma"]
    assert mf["bound"] == "mfma" and abs(mf["frac"] - mf["achieved"] / mf["peak"]) < 1e-3


def test_bench_line_carries_the_parity_gates_of_survey_8d():
    """Round 6 on: the line holds `parity` — the timed pipeline against the CPU oracle on the cpu_baseline frames (SURVEY.md §8d
    "parity gates run with every benchmark") — and every gate is green in the committed line."""
    path = _latest()
    if int(re.match(r"r(\d+)_", os.path.basename(path)).group(1)) < 6:
        return
    p = json.load(open(path))["parity"]
    for k in ("ok", "frames", "voxel_exact", "pillar_rel", "canvas_rel", "feat2d_rel", "box_rel", "nms_exact_on_gpu_logits", "survivors_common",
              "survivors_common_same_score_order", "survivor_flips", "survivor_flips_unexplained", "rtol"):
        assert k in p, k
    assert p["ok"] is True and p["frames"] >= 10 and p["voxel_exact"] is True and p["nms_exact_on_gpu_logits"] is True
    assert max(p["pillar_rel"], p["canvas_rel"], p["feat2d_rel"], p["box_rel"]) <= p["rtol"] == 1e-3
    assert p["survivor_flips_unexplained"] == 0
    a, b = p["survivors_common_same_score_order"]
    assert a >= 0.99 * b > 0
